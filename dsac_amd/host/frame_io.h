// frame_io.h -- data side of the driver programs: 7-Scenes pose files in and out, and the frames the engine consumes.
//
// The reference's programs read RGB-D frames through jp::Dataset and turn them into scene coordinates with the first CNN
// (core/dataset.h, getCoordImg core/cnn_softam.h:211-281); neither the dataset nor the Torch weights exist here, so a frame of these
// drivers IS the predicted scene-coordinate map (the output of that CNN) plus the ground-truth pose file of the dataset:
//     <split>/<scene>/coords/<stem>.coords   "DSACCRD1", int32 H, W, hasSampling, float32 xyz[H*W*3] (mm), [float32 sampling[H*W*2]]
//     <split>/<scene>/poses/<stem>.txt       4 x 4 camera-to-world matrix in metres, the 7-Scenes convention (core/read_data.cpp:69-133)
//     <split>/<scene>/replay/<stem>.sets     optional: N lines "p0 p1 p2 p3" (pixel indices y*W+x) = the minimal sets to evaluate
//     <split>/<scene>/replay/<stem>.perm     optional: int32 steps, P, then steps*P pixel indices = the refinement permutations
// (the two replay files make a run comparable with a recorded reference run that drew its own random numbers);
// ./translation.txt in the working directory is honoured exactly as the reference does.  Files pair up in sorted order.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "cnn_softam.h"

namespace dsac {

struct DriverFrame {
    std::string name;
    int H = 0, W = 0;
    std::vector<float> estObj;     // H*W*3, mm
    std::vector<float> sampling;   // H*W*2 or empty (implicit full-resolution grid)
    Hypothesis poseGT;
    bool havePose = false;
    std::vector<std::array<int32_t, 4>> sets;  // empty: drawn by the engine
    std::vector<int32_t> pixelIdxs;            // empty: dsac::refinePermutations
    int permSteps = 0;
};

// jp::readData(infoFile, info) + Hypothesis(info): core/read_data.cpp:69-133, core/Hypothesis.cpp:45-58.  float arithmetic where the
// reference uses cv::Mat_<float>.  false (and an identity pose) when the file cannot be opened.
bool readPose7Scenes(const std::string& infoFile, Hypothesis& out);
// The inverse, for writing test fixtures and results: pose -> the 4 x 4 matrix a pose file holds (before translation.txt is added back).
std::array<double, 16> poseTo7ScenesMatrix(const Hypothesis& h);
// "convert back to 7-Scenes norm" of core/test_ransac_softam.cpp:161-210: Rodrigues vector + translation in metres, translation.txt added.
std::vector<double> exportPose7Scenes(const cv_trans_t& refAvgHyp);

std::vector<std::string> getSubPaths(const std::string& basePath);  // core/util.cpp: sorted sub-directories
// all frames of the first scene below `splitDir` ("./test/", "./training/"); throws dsac::Error when there is none
std::vector<DriverFrame> loadFrames(const std::string& splitDir);
bool writeCoordsFile(const std::string& path, const DriverFrame& f);
// "chess"-like synthetic frame (stand-in for the CNN's prediction on a 7-Scenes frame): 70 % inliers with 20 mm noise, 30 % outliers
DriverFrame synthFrame(int H, int W, const Camera& cam, unsigned long long seed);

// cv::meanStdDev of a vector (population standard deviation) and the reference's median (element size/2 of the sorted vector)
void meanStdDev(const std::vector<double>& v, double& mean, double& stddev);
double medianOf(std::vector<double> v);

}  // namespace dsac

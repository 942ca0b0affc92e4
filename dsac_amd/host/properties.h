// properties.h -- the parameter set of the reference's programs: names, defaults, `default.config` and `-key value` parsing of
// core/properties.{h,cpp} (GlobalProperties: defaults :39-72, readArguments :97-266, parseCmdLine :269-274, parseConfig :277-306,
// getCamMat :308-323).  Same keys, same defaults, same precedence (config file first, command line second, unknown key stops the
// parse with "unkown argument" exactly as the reference does).  Keys the reference does not have are listed separately below: they
// configure the data source and the soft-inlier score that stand where the reference has its two Torch CNNs.
#pragma once
#include <string>
#include <vector>

#include "cnn_softam.h"

namespace dsac {

struct PoseParameters {               // core/properties.h  struct PoseParameters
    bool randomDraw = true;           // -rdraw
    int ransacIterations = 256;       // -rI    number of hypotheses
    int ransacRefinementIterations = 8;  // -rRI
    int ransacBatchSize = 100;        // -rB    inliers per refinement step
    float ransacSubSample = 0.01f;    // -rSS   ratio of pixels for which refinement gradients are calculated
    float ransacInlierThreshold2D = 10;   // -rT2D (truncated to int by the programs, test_ransac_softam.cpp:51)
    float ransacInlierThreshold3D = 100;  // -rT3D
};

struct DatasetParameters {            // core/properties.h  struct DatasetParameters
    bool rawData = true;              // -rd
    float focalLength = 525;          // -fl
    float xShift = 0, yShift = 0;     // -xs -ys
    float secondaryFocalLength = 585; // -sfl
    float rawXShift = 0, rawYShift = 0;  // -rxs -rys
    int imageWidth = 640, imageHeight = 480;  // -iw -ih
    std::string objScript = "train_obj.lua", scoreScript = "train_score.lua";            // -oscript -sscript
    std::string objModel = "obj_model_init.net", scoreModel = "score_model_init.net";    // -omodel -smodel
    std::string config = "default";
};

// Not in the reference: what replaces its CNN stages and its 7-Scenes reader in these drivers.
struct EngineParameters {
    int synthFrames = 0;      // -synth K   : K synthetic "chess"-like frames instead of ./test/ or ./training/
    int mapWidth = 0, mapHeight = 0;  // -mw -mh : size of synthetic coordinate maps (default 40 x 40 as the reference's sub-sampling)
    unsigned long long seed = 1305;   // -seed   : sampling seed (the reference's ThreadRand seed, thread_rand.h:100)
    float tau = 10.f, beta = 0.5f;    // -tau -beta : soft-inlier score sigmoid (score-CNN stand-in, cnn_softam.h:1072)
    double alpha = 0.1;               // -alpha  : score scale before the softmax
    int rounds = 0;                   // -rounds : training rounds (reference: 5000, train_ransac_softam.cpp:49); 0 = that default
    int device = 0;                   // -dev
    bool indexQuirk = false;          // -quirk  : reproduce the transposed pixel index of path II (cnn_softam.h:628,641)
    int batch = 16;                   // -batch  : images per launch chain of the evaluation program (FrameBatch); 0 = one image per call (Frame::processImage).
                                      //           Training program: frames per round, device-resident (default there: 0 = the reference's per-image loop)
    bool batchGiven = false;          //           (set when -batch is on the command line)
    int refstream = 0;                // -refstream T : draw the minimal sets from the reference's own generators, std::mt19937(seed + t) of T OpenMP threads (core/thread_rand.cpp:40-69;
                                      //           dsac_sample_refstream): the evaluation program's sets equal the reference's for the same seed and thread count, no <stem>.sets
                                      //           replay file needed; frames with a stochastic sub-sampling grid skip the 6400 outputs it drew (-refsub 0: do not)
    int refsub = 1;
    int passes = 1;                   // -passes : process the data set this many times (the first pass warms the device up; timing is reported per pass)
    bool errorImages = true;          // -errimg : write the N error images of every image (the score CNN's input) as the reference does
    int warmupMs = 0;                 // -warmup : run untimed (and unlogged) passes / rounds for this many milliseconds first -- a GPU that idled while the host made
                                      //           or loaded the data set needs ~0.3 s of work before its clock has settled (batch paths only)
    int gradStats = 1;                // -gradstats : the training program prints / logs the gradient statistics every this many rounds (0: never; -batch only)
    bool seam = false;                // -seam   : 1 = the score comes from OUTSIDE the library through the score-CNN seam of the batch path (FrameBatch::processImages /
                                      //           backward with a ScoreModel: error images out, scores in, score gradients out, gradient images in -- cnn_softam.h:1066-1078,
                                      //           train_ransac_softam.cpp:378-383); the model of these programs is the soft-inlier score dressed as an external one
    int defer = 2;                    // -defer  : 0 batches in stream order, 1 refinement tail of a batch under the next batch, 2 score tail too (dsac_hip.h "pi_defer_tail")
};

class GlobalProperties {
public:
    PoseParameters pP;
    DatasetParameters dP;
    EngineParameters eP;
    static GlobalProperties* getInstance();
    bool readArguments(std::vector<std::string> argv);  // false on an unknown key (and the rest is ignored), as the reference
    void parseCmdLine(int argc, const char* argv[]);
    void parseConfig();                                  // reads "<dP.config>.config" from the working directory when present
    Camera getCamMat() const;                            // f = focalLength for both axes, c = (iw/2 + xs, ih/2 + ys) with INTEGER halves
private:
    GlobalProperties() = default;
};

std::vector<std::string> split(const std::string& s);   // core/util.cpp: whitespace split
std::string intToString(int number, int minLength = 0);  // core/util.cpp

}  // namespace dsac

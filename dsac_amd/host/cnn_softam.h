// cnn_softam.h -- C++ host shim: the reference's soft-argmax function surface (core/cnn_softam.h, core/maxloss.h)
// on top of the C ABI of libdsac_hip.so (include/dsac_hip.h).  Same names and argument meaning as the reference;
// cv::Mat / std::vector<cv::Point..> containers are replaced by std::vector PODs, and the (estObj, sampling,
// camMat) triple that every reference function takes is bound once in a `Frame`.
//
//   reference (file:line under /root/reference/core)            here
//   ------------------------------------------------------------------------------------------------------
//   getDiffMap            cnn_softam.h:319                       Frame::getDiffMap / getDiffMaps (batched)
//   softMax / entropy     cnn_softam.h:535 / :80                 softMax / entropy (Frame::softArgMax on the GPU)
//   dPNP                  cnn_softam.h:101                       Frame::dPNP
//   dScore                cnn_softam.h:564                       Frame::dScore
//   refine                cnn_softam.h:663                       Frame::refine
//   dRefineHyp/dRefineObj cnn_softam.h:738 / :853                Frame::dRefine (one batched launch)
//   maxLoss / dLossMax    maxloss.h:69 / :87                     Frame::maxLoss / dLossMax
//   processImage          cnn_softam.h:960                       Frame::processImage
//
// Every call throws dsac::Error (carrying the dsac_status and dsac_last_error text) on failure; nothing is
// computed on the CPU here -- without a gfx950 device the Frame constructor throws.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dsac_hip.h"
#include "Hypothesis.h"

#define CNN_OBJ_MAXINPUT 100.0  // core/lua_calls.h:36

namespace dsac {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Camera { float fx = 525.f, fy = 525.f, cx = 320.f, cy = 240.f; };  // core/properties.cpp:308-323

struct ProcessImageResult {  // the output parameters of processImage, cnn_softam.h:971-988
    std::vector<cv_trans_t> hyps;
    std::vector<std::array<int32_t, 4>> imgIdx;  // minimal sets as pixel indices y*W+x
    std::vector<double> sfScores;
    double sfEntropy = 0;
    cv_trans_t avgHyp, refAvgHyp;
    std::vector<int32_t> inlierMap;
    int refStepsDone = 0;
    double loss = 0, rotErr = 0, tErr = 0;
    bool correct = false;
};

class Frame {
public:
    // estObj: H*W*3 float32 mm; sampling: H*W*2 float32 (u,v) or nullptr for the full-resolution grid.
    Frame(const float* estObj, const float* sampling, int H, int W, const Camera& cam, int device = 0, bool quantiseInt16 = false);
    ~Frame();
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;

    int rows() const { return H_; }
    int cols() const { return W_; }

    // sampling loop of processImage (cnn_softam.h:1010-1060); returns the per-hypothesis success flags
    std::vector<uint8_t> sampleHypotheses(int objHyps, uint64_t seed, int inlierThreshold2D, std::vector<cv_trans_t>& hyps,
                                          std::vector<std::array<int32_t, 4>>& imgIdx, int maxTries = 1 << 20);
    std::vector<float> getDiffMap(const cv_trans_t& hyp);                            // H*W floats
    std::vector<float> getDiffMaps(const std::vector<cv_trans_t>& hyps);             // N*H*W floats
    std::vector<double> softInlierScores(const std::vector<cv_trans_t>& hyps, float tau, float beta);
    // softMax + entropy + soft-argmax average on the GPU
    std::vector<double> softArgMax(const std::vector<double>& scores, double scale, const std::vector<cv_trans_t>& hyps, double& sfEntropy,
                                   cv_trans_t& avgHyp);
    std::vector<double> dPNP(const std::vector<std::array<int32_t, 4>>& imgIdx, float eps = 0.1f);  // N x 6 x 12
    // dScore part (iii): accumulates into jacobean (H*W*3 doubles)
    void dScore(const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx, const std::vector<float>& dDiffMaps,
                std::vector<double>& jacobean, bool referenceIndexQuirk = false);
    cv_trans_t refine(int inlierCount, int refSteps, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs, const cv_trans_t& initHyp,
                      std::vector<int32_t>* inlierMap = nullptr, int* stepsDone = nullptr);
    // dRefineHyp (6x6, row-major) and dRefineObj (sparse: pixels + 6x3 blocks)
    void dRefine(int inlierCount, int refSteps, float inlierThreshold2D, float subSampleFactor, const std::vector<int32_t>& pixelIdxs,
                 const cv_trans_t& initHyp, const std::vector<int32_t>& inlierMap, std::array<double, 36>& dRefineHyp,
                 std::vector<int32_t>& objPixels, std::vector<double>& dRefineObj);
    double maxLoss(const Hypothesis& gt, const cv_trans_t& est, double* rotErr = nullptr, double* tErr = nullptr, bool* correct = nullptr);
    std::array<double, 6> dLossMax(const cv_trans_t& est, const Hypothesis& gt);
    // givenSets (optional): evaluate these minimal sets instead of drawing them (replay of a recorded run)
    ProcessImageResult processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                    const std::vector<int32_t>& pixelIdxs, float tau = 10.f, float beta = 0.5f, double alpha = 0.1,
                                    const std::vector<std::array<int32_t, 4>>* givenSets = nullptr);
    // The backward section of the trainer, core/train_ransac_softam.cpp:288-394: dLoss/d(scene coordinates), H*W x 3, from the
    // forward pass's results.  Path I: dLossMax . (dRefineObj + dRefineHyp . sum_h w_h dPNP_h); path II: softmax backward -> score
    // gradients -> (soft-inlier) score backward.  referenceIndexQuirk reproduces the transposed pixel index of path II (:628,641).
    std::vector<double> backward(const ProcessImageResult& fwd, const Hypothesis& poseGT, int inlierThreshold2D, int inlierCount, int refSteps,
                                 float subSampleFactor, const std::vector<int32_t>& pixelIdxs, float tau = 10.f, float beta = 0.5f,
                                 double alpha = 0.1, bool referenceIndexQuirk = false);

    dsac_ctx* context() { return ctx_; }

private:
    void check(int rc, const char* what);
    dsac_ctx* ctx_ = nullptr;
    int H_, W_;
};

// host-side forms of the two trivial reference functions (cnn_softam.h:535-553, :80-88)
std::vector<double> softMax(const std::vector<double>& scores);
double entropy(const std::vector<double>& dist);
// the reference's permutation stream: one default-seeded std::mt19937, Fisher-Yates per step (cnn_softam.h:1104-1114)
std::vector<int32_t> refinePermutations(int P, int refSteps);

}  // namespace dsac

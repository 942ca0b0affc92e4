// cnn.h -- C++ host shim for the DSAC (probabilistic selection) variant: the function surface of core/cnn.h on top of the C
// ABI, as an extension of the soft-argmax shim (cnn_softam.h).  Same conventions: PODs instead of cv::Mat, the
// (estObj, sampling, camMat) triple bound once in the Frame, nothing computed on the CPU except the two trivial sums.
//
//   reference (file:line under /root/reference/core)                 here
//   ----------------------------------------------------------------------------------------------------------
//   draw                  cnn.h:102                                   draw (host; uniform number supplied by the caller)
//   expectedMaxLoss       cnn.h:137                                   DsacFrame::expectedMaxLoss (losses on the GPU, one batch)
//   refinement of all N   cnn.h:1155-1215 (inside processImage)       DsacFrame::refineAll
//   dRefine               cnn.h:854                                   DsacFrame::dRefine (one batched launch per hypothesis)
//   dSMScore              cnn.h:726                                   DsacFrame::dSMScore (row-major result, like the reference's)
//   processImage          cnn.h:1000                                  DsacFrame::processImage
#pragma once
#include "cnn_softam.h"

namespace dsac {

struct ProcessImageDsacResult {  // the output parameters of processImage, core/cnn.h:1011-1029
    std::vector<cv_trans_t> hyps, refHyps;
    std::vector<std::array<int32_t, 4>> imgIdx;
    std::vector<double> sfScores, losses;
    std::vector<int32_t> inlierMaps;  // N x H*W, the minimal sets' own cells cleared
    double sfEntropy = 0, expectedLoss = 0, rotErr = 0, tErr = 0;
    int hypIdx = 0;
    bool correct = false;
};

// draw (cnn.h:102-127): entries below EPS skipped; u in [0,1) picks along the cumulative distribution, u < 0 = the most probable entry
int draw(const std::vector<double>& probs, double u);

class DsacFrame : public Frame {
public:
    using Frame::Frame;
    std::vector<cv_trans_t> refineAll(int inlierCount, int refSteps, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs,
                                      const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx,
                                      std::vector<int32_t>* inlierMaps = nullptr);
    double expectedMaxLoss(const Hypothesis& gt, const std::vector<cv_trans_t>& hyps, const std::vector<double>& probs, std::vector<double>& losses,
                           std::vector<std::array<double, 6>>* dLosses = nullptr);
    // 6 x 9 block of the first three set points (columns pt*3 + c) and the sparse inlier-cell part as in Frame::dRefine
    void dRefine(int inlierCount, int refSteps, float subSampleFactor, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs,
                 const std::array<int32_t, 4>& imgIdx, const int32_t* inlierMap, std::array<double, 54>& dRefineSet, std::vector<int32_t>& objPixels,
                 std::vector<double>& dRefineObj);
    // sum over hypotheses of dSMScore's per-hypothesis Jacobians (train_ransac.cpp:362-368), accumulated into jacobean (H*W*3)
    void dSMScore(const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx, const std::vector<double>& losses,
                  const std::vector<double>& sfScores, const std::vector<float>& dDiffMaps, std::vector<double>& jacobean,
                  std::vector<double>* scoreOutputGradients = nullptr);
    ProcessImageDsacResult processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                        const std::vector<int32_t>& pixelIdxs, double drawU = -1, float tau = 10.f, float beta = 0.5f, double alpha = 0.1);
};

}  // namespace dsac

// cnn.cpp -- see cnn.h.  Marshals std::vector containers into the C ABI; no geometry on the CPU.
#include "cnn.h"

#include <cmath>

namespace dsac {

static std::vector<double> flatten_poses(const std::vector<cv_trans_t>& h) {
    std::vector<double> v(h.size() * 6);
    for (size_t i = 0; i < h.size(); i++) {
        const Pose6 p = pack(h[i]);
        for (int k = 0; k < 6; k++) v[i * 6 + k] = p[k];
    }
    return v;
}
static void dcheck(dsac_ctx* c, int rc, const char* what) {
    if (rc != DSAC_OK) throw Error(rc, std::string(what) + ": " + dsac_last_error(c));
}

int draw(const std::vector<double>& probs, double u) {
    const double EPS = 0.00000001;  // core/types.h:32
    double sum = 0, best = -1;
    int bestIdx = 0;
    for (size_t i = 0; i < probs.size(); i++) {
        if (probs[i] < EPS) continue;
        sum += probs[i];
        if (best < 0 || probs[i] > best) { best = probs[i]; bestIdx = (int)i; }
    }
    if (u < 0) return bestIdx;
    const double r = u * sum;  // drand(0, probSum), cnn.h:123
    double cum = 0;
    int last = bestIdx;
    for (size_t i = 0; i < probs.size(); i++) {
        if (probs[i] < EPS) continue;
        cum += probs[i];
        last = (int)i;
        if (cum > r) return (int)i;  // std::map::upper_bound: first cumulative value greater than r
    }
    return last;
}

std::vector<cv_trans_t> DsacFrame::refineAll(int inlierCount, int refSteps, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs,
                                             const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx,
                                             std::vector<int32_t>* inlierMaps) {
    const int N = (int)hyps.size();
    const std::vector<double> in = flatten_poses(hyps);
    std::vector<double> out((size_t)N * 6);
    if (inlierMaps) inlierMaps->assign((size_t)N * rows() * cols(), 0);
    dcheck(context(), dsac_refine_all(context(), N, in.data(), pixelIdxs.data(), refSteps, inlierCount, 50, inlierThreshold2D,
                                      imgIdx.empty() ? nullptr : imgIdx[0].data(), out.data(), inlierMaps ? inlierMaps->data() : nullptr, nullptr),
           "dsac_refine_all");
    std::vector<cv_trans_t> r(N);
    for (int h = 0; h < N; h++) r[h] = unpack(Pose6{out[(size_t)h * 6], out[(size_t)h * 6 + 1], out[(size_t)h * 6 + 2], out[(size_t)h * 6 + 3], out[(size_t)h * 6 + 4], out[(size_t)h * 6 + 5]});
    return r;
}

double DsacFrame::expectedMaxLoss(const Hypothesis& gt, const std::vector<cv_trans_t>& hyps, const std::vector<double>& probs, std::vector<double>& losses,
                                  std::vector<std::array<double, 6>>* dLosses) {
    const int N = (int)hyps.size();
    const std::vector<double> in = flatten_poses(hyps), g = gt.getRodVecAndTrans();
    std::vector<double> out4((size_t)N * 4), J(dLosses ? (size_t)N * 6 : 0);
    dcheck(context(), dsac_loss_batch(context(), N, in.data(), g.data(), out4.data(), dLosses ? J.data() : nullptr), "dsac_loss_batch");
    losses.resize(N);
    double loss = 0;
    for (int i = 0; i < N; i++) { losses[i] = out4[(size_t)i * 4]; loss += probs[i] * losses[i]; }
    if (dLosses) {
        dLosses->resize(N);
        for (int i = 0; i < N; i++) for (int k = 0; k < 6; k++) (*dLosses)[i][k] = J[(size_t)i * 6 + k];
    }
    return loss;
}

void DsacFrame::dRefine(int inlierCount, int refSteps, float subSampleFactor, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs,
                        const std::array<int32_t, 4>& imgIdx, const int32_t* inlierMap, std::array<double, 54>& dRefineSet, std::vector<int32_t>& objPixels,
                        std::vector<double>& dRefineObj) {
    const int cap = 4096;
    objPixels.assign(cap, 0);
    dRefineObj.assign((size_t)cap * 18, 0.0);
    int32_t n = 0;
    dcheck(context(), dsac_refine_fd_set(context(), imgIdx.data(), pixelIdxs.data(), refSteps, inlierCount, 50, inlierThreshold2D, inlierMap, subSampleFactor,
                                         2.f, dRefineSet.data(), objPixels.data(), dRefineObj.data(), cap, &n),
           "dsac_refine_fd_set");
    objPixels.resize(n);
    dRefineObj.resize((size_t)n * 18);
}

void DsacFrame::dSMScore(const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx, const std::vector<double>& losses,
                         const std::vector<double>& sfScores, const std::vector<float>& dDiffMaps, std::vector<double>& jacobean,
                         std::vector<double>* scoreOutputGradients) {
    if (scoreOutputGradients) {  // cnn.h:737-742, O(N) form
        double mean = 0;
        for (size_t j = 0; j < sfScores.size(); j++) mean += sfScores[j] * losses[j];
        scoreOutputGradients->resize(sfScores.size());
        for (size_t i = 0; i < sfScores.size(); i++) (*scoreOutputGradients)[i] = sfScores[i] * (losses[i] - mean);
    }
    // the reference re-orders dScore's blocks to row-major (cnn.h:749-765): no index quirk
    dScore(hyps, imgIdx, dDiffMaps, jacobean, /*referenceIndexQuirk=*/false);
}

ProcessImageDsacResult DsacFrame::processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                               const std::vector<int32_t>& pixelIdxs, double drawU, float tau, float beta, double alpha) {
    ProcessImageDsacResult r;
    sampleHypotheses(objHyps, seed, inlierThreshold2D, r.hyps, r.imgIdx);
    const std::vector<double> scores = softInlierScores(r.hyps, tau, beta);  // the score-CNN seam of cnn.h:1134
    cv_trans_t unusedAvg;
    r.sfScores = softArgMax(scores, alpha, r.hyps, r.sfEntropy, unusedAvg);
    r.hypIdx = draw(r.sfScores, drawU);
    r.refHyps = refineAll(inlierCount, refSteps, (float)inlierThreshold2D, pixelIdxs, r.hyps, r.imgIdx, &r.inlierMaps);
    r.expectedLoss = expectedMaxLoss(poseGT, r.refHyps, r.sfScores, r.losses);
    maxLoss(poseGT, r.refHyps[r.hypIdx], &r.rotErr, &r.tErr, &r.correct);
    return r;
}

}  // namespace dsac

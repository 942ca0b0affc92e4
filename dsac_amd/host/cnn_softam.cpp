// cnn_softam.cpp -- see cnn_softam.h.  Marshals std::vector containers into the C ABI; no geometry on the CPU.
#include "cnn_softam.h"

#include <cmath>
#include <random>
#include <stdexcept>

namespace dsac {

static std::vector<double> flatten(const std::vector<cv_trans_t>& h) {
    std::vector<double> v(h.size() * 6);
    for (size_t i = 0; i < h.size(); i++) {
        const Pose6 p = pack(h[i]);
        for (int k = 0; k < 6; k++) v[i * 6 + k] = p[k];
    }
    return v;
}

void Frame::check(int rc, const char* what) {
    if (rc != DSAC_OK) throw Error(rc, std::string(what) + ": " + dsac_last_error(ctx_));
}

Frame::Frame(const float* estObj, const float* sampling, int H, int W, const Camera& cam, int device, bool quantiseInt16) : H_(H), W_(W) {
    int rc = dsac_create(&ctx_, device);
    if (rc != DSAC_OK) throw Error(rc, std::string("dsac_create: ") + dsac_last_error(nullptr));
    rc = dsac_set_frame(ctx_, estObj, sampling, H, W, cam.fx, cam.fy, cam.cx, cam.cy, quantiseInt16 ? DSAC_FRAME_QUANTISE_INT16 : 0u);
    if (rc != DSAC_OK) {
        const std::string msg = dsac_last_error(ctx_);
        dsac_destroy(ctx_);
        ctx_ = nullptr;
        throw Error(rc, "dsac_set_frame: " + msg);
    }
}

Frame::~Frame() { dsac_destroy(ctx_); }

std::vector<uint8_t> Frame::sampleHypotheses(int objHyps, uint64_t seed, int inlierThreshold2D, std::vector<cv_trans_t>& hyps,
                                             std::vector<std::array<int32_t, 4>>& imgIdx, int maxTries) {
    std::vector<double> poses((size_t)objHyps * 6);
    std::vector<uint8_t> ok(objHyps);
    imgIdx.assign(objHyps, {0, 0, 0, 0});
    check(dsac_sample(ctx_, objHyps, seed, nullptr, (float)inlierThreshold2D, maxTries, poses.data(), &imgIdx[0][0], ok.data()), "dsac_sample");
    hyps.resize(objHyps);
    for (int h = 0; h < objHyps; h++) hyps[h] = unpack({poses[h * 6], poses[h * 6 + 1], poses[h * 6 + 2], poses[h * 6 + 3], poses[h * 6 + 4], poses[h * 6 + 5]});
    return ok;
}

std::vector<float> Frame::getDiffMaps(const std::vector<cv_trans_t>& hyps) {
    std::vector<float> err(hyps.size() * (size_t)H_ * W_);
    const std::vector<double> p = flatten(hyps);
    check(dsac_reproject(ctx_, (int)hyps.size(), p.data(), (float)CNN_OBJ_MAXINPUT, err.data(), 0.f, 0.f, nullptr), "dsac_reproject");
    return err;
}

std::vector<float> Frame::getDiffMap(const cv_trans_t& hyp) { return getDiffMaps({hyp}); }

std::vector<double> Frame::softInlierScores(const std::vector<cv_trans_t>& hyps, float tau, float beta) {
    std::vector<double> s(hyps.size());
    const std::vector<double> p = flatten(hyps);
    check(dsac_reproject(ctx_, (int)hyps.size(), p.data(), (float)CNN_OBJ_MAXINPUT, nullptr, tau, beta, s.data()), "dsac_reproject");
    return s;
}

std::vector<double> Frame::softArgMax(const std::vector<double>& scores, double scale, const std::vector<cv_trans_t>& hyps, double& sfEntropy,
                                      cv_trans_t& avgHyp) {
    std::vector<double> w(scores.size());
    const std::vector<double> p = flatten(hyps);
    double avg[6];
    check(dsac_softmax(ctx_, (int)scores.size(), scores.data(), scale, w.data(), &sfEntropy, p.data(), avg), "dsac_softmax");
    avgHyp = unpack({avg[0], avg[1], avg[2], avg[3], avg[4], avg[5]});
    return w;
}

std::vector<double> Frame::dPNP(const std::vector<std::array<int32_t, 4>>& imgIdx, float eps) {
    std::vector<double> J(imgIdx.size() * 72);
    check(dsac_dpnp(ctx_, (int)imgIdx.size(), &imgIdx[0][0], eps, J.data()), "dsac_dpnp");
    return J;
}

void Frame::dScore(const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx, const std::vector<float>& dDiffMaps,
                   std::vector<double>& jacobean, bool referenceIndexQuirk) {
    jacobean.resize((size_t)H_ * W_ * 3, 0.0);
    const std::vector<double> p = flatten(hyps);
    check(dsac_score_backward(ctx_, (int)hyps.size(), p.data(), &imgIdx[0][0], dDiffMaps.data(), nullptr,
                              referenceIndexQuirk ? DSAC_BWD_QUIRK_TRANSPOSE : 0u, jacobean.data()),
          "dsac_score_backward");
}

cv_trans_t Frame::refine(int inlierCount, int refSteps, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs, const cv_trans_t& initHyp,
                         std::vector<int32_t>* inlierMap, int* stepsDone) {
    const Pose6 in = pack(initHyp);
    double out[6];
    int32_t sd = 0;
    if (inlierMap) inlierMap->assign((size_t)H_ * W_, 0);
    check(dsac_refine(ctx_, 1, in.data(), pixelIdxs.data(), refSteps, inlierCount, 50, inlierThreshold2D, nullptr, nullptr, out,
                      inlierMap ? inlierMap->data() : nullptr, &sd),
          "dsac_refine");
    if (stepsDone) *stepsDone = sd;
    return unpack({out[0], out[1], out[2], out[3], out[4], out[5]});
}

void Frame::dRefine(int inlierCount, int refSteps, float inlierThreshold2D, float subSampleFactor, const std::vector<int32_t>& pixelIdxs,
                    const cv_trans_t& initHyp, const std::vector<int32_t>& inlierMap, std::array<double, 36>& dRefineHyp,
                    std::vector<int32_t>& objPixels, std::vector<double>& dRefineObj) {
    const Pose6 in = pack(initHyp);
    const int cap = 4096;
    objPixels.assign(cap, 0);
    dRefineObj.assign((size_t)cap * 18, 0.0);
    int32_t n = 0;
    check(dsac_refine_fd(ctx_, in.data(), pixelIdxs.data(), refSteps, inlierCount, 50, inlierThreshold2D, inlierMap.data(), subSampleFactor, 0.001f,
                         2.f, dRefineHyp.data(), objPixels.data(), dRefineObj.data(), cap, &n),
          "dsac_refine_fd");
    objPixels.resize(n);
    dRefineObj.resize((size_t)n * 18);
}

double Frame::maxLoss(const Hypothesis& gt, const cv_trans_t& est, double* rotErr, double* tErr, bool* correct) {
    const Pose6 e = pack(est);
    const std::vector<double> g = gt.getRodVecAndTrans();
    double out4[4];
    check(dsac_loss(ctx_, e.data(), g.data(), out4, nullptr), "dsac_loss");
    if (rotErr) *rotErr = out4[1];
    if (tErr) *tErr = out4[2];
    if (correct) *correct = out4[3] > 0.5;
    return out4[0];
}

std::array<double, 6> Frame::dLossMax(const cv_trans_t& est, const Hypothesis& gt) {
    const Pose6 e = pack(est);
    const std::vector<double> g = gt.getRodVecAndTrans();
    std::array<double, 6> J{};
    double out4[4];
    check(dsac_loss(ctx_, e.data(), g.data(), out4, J.data()), "dsac_loss");
    return J;
}

ProcessImageResult Frame::processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                       const std::vector<int32_t>& pixelIdxs, float tau, float beta, double alpha,
                                       const std::vector<std::array<int32_t, 4>>* givenSets) {
    ProcessImageResult r;
    if (givenSets && !givenSets->empty()) {
        objHyps = (int)givenSets->size();
        std::vector<double> poses((size_t)objHyps * 6);
        std::vector<uint8_t> ok(objHyps);
        r.imgIdx.assign(objHyps, {0, 0, 0, 0});
        check(dsac_sample(ctx_, objHyps, seed, &(*givenSets)[0][0], (float)inlierThreshold2D, 1, poses.data(), &r.imgIdx[0][0], ok.data()), "dsac_sample");
        r.hyps.resize(objHyps);
        for (int h = 0; h < objHyps; h++) r.hyps[h] = unpack({poses[h * 6], poses[h * 6 + 1], poses[h * 6 + 2], poses[h * 6 + 3], poses[h * 6 + 4], poses[h * 6 + 5]});
    } else {
        sampleHypotheses(objHyps, seed, inlierThreshold2D, r.hyps, r.imgIdx);
    }
    const std::vector<double> scores = softInlierScores(r.hyps, tau, beta);  // the score-CNN seam of cnn_softam.h:1072
    r.sfScores = softArgMax(scores, alpha, r.hyps, r.sfEntropy, r.avgHyp);
    r.refAvgHyp = refine(inlierCount, refSteps, (float)inlierThreshold2D, pixelIdxs, r.avgHyp, &r.inlierMap, &r.refStepsDone);
    r.loss = maxLoss(poseGT, r.refAvgHyp, &r.rotErr, &r.tErr, &r.correct);
    return r;
}

std::vector<double> Frame::backward(const ProcessImageResult& fwd, const Hypothesis& poseGT, int inlierThreshold2D, int inlierCount, int refSteps,
                                    float subSampleFactor, const std::vector<int32_t>& pixelIdxs, float tau, float beta, double alpha,
                                    bool referenceIndexQuirk) {
    const size_t P = (size_t)H_ * W_;
    const int N = (int)fwd.hyps.size();
    std::vector<double> grad(P * 3, 0.0);
    // --- path I and the softmax backward (train_ransac_softam.cpp:294-376) as one device-side chain: dLossMax at the refined pose,
    // dRefineObj / dRefineHyp (12 + 6n finite-difference replicas in one launch), their contraction with dL, dPNP of every minimal set,
    // the scatter of v6 . w_h dPNP_h to the support points and the softmax backward
    const std::vector<double> p = flatten(fwd.hyps);
    const Pose6 avg = pack(fwd.avgHyp), ref = pack(fwd.refAvgHyp);
    const std::vector<double> gt = poseGT.getRodVecAndTrans();
    std::vector<double> J((size_t)N * 72), g(N);
    // the gradient must belong to the function the forward evaluated: the same refSteps steps of the same permutations (a replayed .perm file
    // may hold more of them than the forward used)
    if (refSteps < 0 || (size_t)refSteps * P > pixelIdxs.size())
        throw std::invalid_argument("Frame::backward: pixelIdxs holds fewer than refSteps permutations");
    if (fwd.refStepsDone > refSteps) throw std::invalid_argument("Frame::backward: the forward pass refined more steps than refSteps");
    const int steps = refSteps;
    check(dsac_backward_path1(ctx_, N, p.data(), &fwd.imgIdx[0][0], fwd.sfScores.data(), avg.data(), ref.data(), gt.data(), pixelIdxs.data(), steps, inlierCount,
                              50, (float)inlierThreshold2D, fwd.inlierMap.data(), subSampleFactor, 0.001f, 2.f, 1.0, J.data(), grad.data(), g.data(), nullptr,
                              nullptr),
          "dsac_backward_path1");
    // --- path II: score gradients -> error images -> object coordinates (dScore, :379-383); the score is alpha * soft-inlier count
    for (double& x : g) x *= alpha;
    if (!referenceIndexQuirk) {
        check(dsac_soft_score_backward(ctx_, N, p.data(), &fwd.imgIdx[0][0], g.data(), (float)CNN_OBJ_MAXINPUT, tau, beta, J.data(), 0u, grad.data()),
              "dsac_soft_score_backward");
        return grad;
    }
    // Parity mode for square (reference-sized) maps: the reference reads the score CNN's input gradient back x-major
    // (core/lua_calls.h:329-335) and dScore then addresses pixel (y, x) as x*cols*3 + y*3 (cnn_softam.h:628,641).  Reproduced literally:
    // the gradient images are formed explicitly, handed over transposed, and K4 writes to the transposed index.
    const std::vector<float> err = getDiffMaps(fwd.hyps);
    std::vector<float> dDiff((size_t)N * P);
    for (int h = 0; h < N; h++)
        for (int y = 0; y < H_; y++)
            for (int x = 0; x < W_; x++) {
                const double e = err[(size_t)h * P + (size_t)x * W_ + y];  // natural gradient of cell (x, y) lands in cell (y, x)
                const double sg = 1.0 / (1.0 + std::exp(-(double)beta * ((double)tau - e)));
                dDiff[(size_t)h * P + (size_t)y * W_ + x] = (float)(g[h] * (-(double)beta) * sg * (1.0 - sg));
            }
    check(dsac_score_backward(ctx_, N, p.data(), &fwd.imgIdx[0][0], dDiff.data(), J.data(), DSAC_BWD_QUIRK_TRANSPOSE, grad.data()),
          "dsac_score_backward");
    return grad;
}

std::vector<double> softMax(const std::vector<double>& scores) {
    double m = 0;
    for (size_t i = 0; i < scores.size(); i++)
        if (i == 0 || scores[i] > m) m = scores[i];
    std::vector<double> sf(scores.size());
    double sum = 0;
    for (size_t i = 0; i < scores.size(); i++) { sf[i] = std::exp(scores[i] - m); sum += sf[i]; }
    for (double& v : sf) v /= sum;
    return sf;
}

double entropy(const std::vector<double>& dist) {
    double e = 0;
    for (double d : dist)
        if (d > 0) e -= d * std::log2(d);
    return e;
}

std::vector<int32_t> refinePermutations(int P, int refSteps) {
    std::mt19937 randG;  // default seed 5489, carried across steps
    std::vector<int32_t> out((size_t)refSteps * P);
    for (int s = 0; s < refSteps; s++) {
        int32_t* a = out.data() + (size_t)s * P;
        for (int i = 0; i < P; i++) a[i] = i;
        for (int i = 1; i < P; i++) {  // libstdc++ 4.8 std::shuffle: swap i with uniform[0, i], rejection down-scaling
            const uint32_t uerange = (uint32_t)i + 1, scaling = 0xFFFFFFFFu / uerange, past = uerange * scaling;
            uint32_t rnd;
            do rnd = (uint32_t)randG(); while (rnd >= past);
            std::swap(a[i], a[rnd / scaling]);
        }
    }
    return out;
}

}  // namespace dsac

// cnn_softam.cpp -- see cnn_softam.h.  Marshals std::vector containers into the C ABI; no geometry on the CPU.
#include "cnn_softam.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
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

// ---- Context -------------------------------------------------------------------------------------------------------------------------
Context::Context(int device) {
    const int rc = dsac_create(&ctx_, device);
    if (rc != DSAC_OK) throw Error(rc, std::string("dsac_create: ") + dsac_last_error(nullptr));
}

Context::~Context() { dsac_destroy(ctx_); }

Context& Context::shared(int device) {
    static std::map<int, std::unique_ptr<Context>> contexts;  // destroyed at exit, after every Frame of a well-formed program
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    std::unique_ptr<Context>& c = contexts[device];
    if (!c) c.reset(new Context(device));
    return *c;
}

void Context::check(int rc, const char* what) {
    if (rc != DSAC_OK) throw Error(rc, std::string(what) + ": " + dsac_last_error(ctx_));
}
void Context::synchronize() { check(dsac_synchronize(ctx_), "dsac_synchronize"); }
void Context::setOption(const char* key, int value) {
    check(dsac_set_option(ctx_, key, value), "dsac_set_option");
    options_[key] = value;
}
int Context::option(const char* key, int unset) const {
    const auto it = options_.find(key);
    return it == options_.end() ? unset : it->second;
}
void* Context::deviceAlloc(size_t bytes) { void* p = nullptr; check(dsac_device_alloc(ctx_, bytes, &p), "dsac_device_alloc"); return p; }
void Context::deviceFree(void* p) noexcept { (void)dsac_device_free(ctx_, p); }
void* Context::hostAlloc(size_t bytes) { void* p = nullptr; check(dsac_host_alloc(ctx_, bytes, &p), "dsac_host_alloc"); return p; }
void Context::hostFree(void* p) noexcept { (void)dsac_host_free(ctx_, p); }
void Context::copy(void* dst, const void* src, size_t bytes) { check(dsac_copy_async(ctx_, dst, src, bytes), "dsac_copy_async"); }
void Context::zero(void* dst, size_t bytes) { check(dsac_fill_zero_async(ctx_, dst, bytes), "dsac_fill_zero_async"); }

// ---- Frame ---------------------------------------------------------------------------------------------------------------------------
void Frame::check(int rc, const char* what) { C_->check(rc, what); }

Frame::Frame(Context& ctx, const float* estObj, const float* sampling, int H, int W, const Camera& cam, bool quantiseInt16)
    : C_(&ctx), ctx_(ctx.get()), H_(H), W_(W), cam_(cam), quantise_(quantiseInt16) {
    if (!estObj || H <= 0 || W <= 0) throw Error(DSAC_ERR_INVALID, "Frame: need estObj and H, W > 0");
    const size_t P = (size_t)H * W;
    xyz_.resize(ctx, P * 3);
    xyz_.upload(estObj, P * 3);
    if (sampling) {
        uv_.resize(ctx, P * 2);
        uv_.upload(sampling, P * 2);
    }
    C_->synchronize();  // the caller's host arrays may go away after the constructor returns
    bind();
}

Frame::Frame(const float* estObj, const float* sampling, int H, int W, const Camera& cam, int device, bool quantiseInt16)
    : Frame(Context::shared(device), estObj, sampling, H, W, cam, quantiseInt16) {}

Frame::~Frame() {
    if (C_ && C_->boundTo() == this) C_->setBound(nullptr);
    // xyz_ / uv_ are freed by their destructors (hipFree waits for work that still reads them)
}

void Frame::bind() {
    if (C_->boundTo() == this) return;
    unsigned flags = DSAC_FRAME_BORROW;
    if (quantise_ && !quantised_) flags |= DSAC_FRAME_QUANTISE_INT16;  // in place, once
    check(dsac_set_frame(ctx_, xyz_.data(), uv_.data(), H_, W_, cam_.fx, cam_.fy, cam_.cx, cam_.cy, flags), "dsac_set_frame");
    quantised_ = quantise_;
    C_->setBound(this);
}

std::vector<uint8_t> Frame::sampleHypotheses(int objHyps, uint64_t seed, int inlierThreshold2D, std::vector<cv_trans_t>& hyps,
                                             std::vector<std::array<int32_t, 4>>& imgIdx, int maxTries) {
    bind();
    std::vector<double> poses((size_t)objHyps * 6);
    std::vector<uint8_t> ok(objHyps);
    imgIdx.assign(objHyps, {0, 0, 0, 0});
    check(dsac_sample(ctx_, objHyps, seed, nullptr, (float)inlierThreshold2D, maxTries, poses.data(), &imgIdx[0][0], ok.data()), "dsac_sample");
    hyps.resize(objHyps);
    for (int h = 0; h < objHyps; h++) hyps[h] = unpack({poses[h * 6], poses[h * 6 + 1], poses[h * 6 + 2], poses[h * 6 + 3], poses[h * 6 + 4], poses[h * 6 + 5]});
    return ok;
}

std::vector<uint8_t> Frame::sampleHypothesesRefStream(int objHyps, int inlierThreshold2D, std::vector<cv_trans_t>& hyps, std::vector<std::array<int32_t, 4>>& imgIdx,
                                                      unsigned long long subSampleOutputs, long long maxAttempts) {
    bind();
    std::vector<double> poses((size_t)objHyps * 6);
    std::vector<uint8_t> ok(objHyps);
    imgIdx.assign(objHyps, {0, 0, 0, 0});
    if (subSampleOutputs) check(dsac_refstream_discard(ctx_, 0, subSampleOutputs), "dsac_refstream_discard");
    check(dsac_sample_refstream(ctx_, objHyps, (float)inlierThreshold2D, maxAttempts, poses.data(), &imgIdx[0][0], ok.data(), nullptr, nullptr), "dsac_sample_refstream");
    hyps.resize(objHyps);
    for (int h = 0; h < objHyps; h++) hyps[h] = unpack({poses[h * 6], poses[h * 6 + 1], poses[h * 6 + 2], poses[h * 6 + 3], poses[h * 6 + 4], poses[h * 6 + 5]});
    return ok;
}

std::vector<float> Frame::getDiffMaps(const std::vector<cv_trans_t>& hyps) {
    bind();
    std::vector<float> err(hyps.size() * (size_t)H_ * W_);
    const std::vector<double> p = flatten(hyps);
    check(dsac_reproject(ctx_, (int)hyps.size(), p.data(), (float)CNN_OBJ_MAXINPUT, err.data(), 0.f, 0.f, nullptr), "dsac_reproject");
    return err;
}

std::vector<float> Frame::getDiffMap(const cv_trans_t& hyp) { return getDiffMaps({hyp}); }

std::vector<double> Frame::softInlierScores(const std::vector<cv_trans_t>& hyps, float tau, float beta) {
    bind();
    std::vector<double> s(hyps.size());
    const std::vector<double> p = flatten(hyps);
    check(dsac_reproject(ctx_, (int)hyps.size(), p.data(), (float)CNN_OBJ_MAXINPUT, nullptr, tau, beta, s.data()), "dsac_reproject");
    return s;
}

std::vector<double> Frame::softArgMax(const std::vector<double>& scores, double scale, const std::vector<cv_trans_t>& hyps, double& sfEntropy,
                                      cv_trans_t& avgHyp) {
    bind();
    std::vector<double> w(scores.size());
    const std::vector<double> p = flatten(hyps);
    double avg[6];
    check(dsac_softmax(ctx_, (int)scores.size(), scores.data(), scale, w.data(), &sfEntropy, p.data(), avg), "dsac_softmax");
    avgHyp = unpack({avg[0], avg[1], avg[2], avg[3], avg[4], avg[5]});
    return w;
}

std::vector<double> Frame::dPNP(const std::vector<std::array<int32_t, 4>>& imgIdx, float eps) {
    bind();
    std::vector<double> J(imgIdx.size() * 72);
    check(dsac_dpnp(ctx_, (int)imgIdx.size(), &imgIdx[0][0], eps, J.data()), "dsac_dpnp");
    return J;
}

void Frame::dScore(const std::vector<cv_trans_t>& hyps, const std::vector<std::array<int32_t, 4>>& imgIdx, const std::vector<float>& dDiffMaps,
                   std::vector<double>& jacobean, bool referenceIndexQuirk) {
    bind();
    jacobean.resize((size_t)H_ * W_ * 3, 0.0);
    const std::vector<double> p = flatten(hyps);
    check(dsac_score_backward(ctx_, (int)hyps.size(), p.data(), &imgIdx[0][0], dDiffMaps.data(), nullptr,
                              referenceIndexQuirk ? DSAC_BWD_QUIRK_TRANSPOSE : 0u, jacobean.data()),
          "dsac_score_backward");
}

cv_trans_t Frame::refine(int inlierCount, int refSteps, float inlierThreshold2D, const std::vector<int32_t>& pixelIdxs, const cv_trans_t& initHyp,
                         std::vector<int32_t>* inlierMap, int* stepsDone) {
    bind();
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
    bind();
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
    bind();
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
    bind();
    const Pose6 e = pack(est);
    const std::vector<double> g = gt.getRodVecAndTrans();
    std::array<double, 6> J{};
    double out4[4];
    check(dsac_loss(ctx_, e.data(), g.data(), out4, J.data()), "dsac_loss");
    return J;
}

void Frame::getDiffMapsDevice(const std::vector<cv_trans_t>& hyps, DeviceArray<float>& out) {
    bind();
    const size_t n = hyps.size() * (size_t)H_ * W_;
    if (out.size() < n) out.resize(*C_, n);
    const std::vector<double> p = flatten(hyps);
    check(dsac_reproject(ctx_, (int)hyps.size(), p.data(), (float)CNN_OBJ_MAXINPUT, out.data(), 0.f, 0.f, nullptr), "dsac_reproject");
}

static cv_trans_t pose_at(const double* v) { return unpack({v[0], v[1], v[2], v[3], v[4], v[5]}); }

ProcessImageResult Frame::processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                       const std::vector<int32_t>& pixelIdxs, float tau, float beta, double alpha,
                                       const std::vector<std::array<int32_t, 4>>* givenSets) {
    bind();
    ProcessImageResult r;
    if (givenSets && !givenSets->empty()) {
        // replay of recorded minimal sets: the stages one by one (dsac_process_images draws its own sets)
        objHyps = (int)givenSets->size();
        std::vector<double> poses((size_t)objHyps * 6);
        std::vector<uint8_t> ok(objHyps);
        r.imgIdx.assign(objHyps, {0, 0, 0, 0});
        check(dsac_sample(ctx_, objHyps, seed, &(*givenSets)[0][0], (float)inlierThreshold2D, 1, poses.data(), &r.imgIdx[0][0], ok.data()), "dsac_sample");
        r.hyps.resize(objHyps);
        for (int h = 0; h < objHyps; h++) r.hyps[h] = pose_at(&poses[(size_t)h * 6]);
        const std::vector<double> scores = softInlierScores(r.hyps, tau, beta);  // the score-CNN seam of cnn_softam.h:1072
        r.sfScores = softArgMax(scores, alpha, r.hyps, r.sfEntropy, r.avgHyp);
        r.refAvgHyp = refine(inlierCount, refSteps, (float)inlierThreshold2D, pixelIdxs, r.avgHyp, &r.inlierMap, &r.refStepsDone);
        r.loss = maxLoss(poseGT, r.refAvgHyp, &r.rotErr, &r.tErr, &r.correct);
        return r;
    }
    if (refSteps < 0 || (size_t)refSteps * H_ * W_ > pixelIdxs.size()) throw std::invalid_argument("Frame::processImage: pixelIdxs holds fewer than refSteps permutations");
    return processImageCall(poseGT, objHyps, seed, inlierThreshold2D, inlierCount, refSteps, pixelIdxs.data(), tau, beta, alpha, true);
}

ProcessImageResult Frame::processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                       const DeviceArray<int32_t>& pixelIdxsDevice, float tau, float beta, double alpha, bool wantInlierMap) {
    if (refSteps < 0 || (size_t)refSteps * H_ * W_ > pixelIdxsDevice.size()) throw std::invalid_argument("Frame::processImage: pixelIdxs holds fewer than refSteps permutations");
    return processImageCall(poseGT, objHyps, seed, inlierThreshold2D, inlierCount, refSteps, pixelIdxsDevice.data(), tau, beta, alpha, wantInlierMap);
}

// the whole of core/cnn_softam.h:960-1179 in ONE call: sample + P3P, soft-inlier scores, softmax / entropy / soft-argmax pose, the refinement loop,
// maxLoss -- one launch chain on the device, one copy-back of the small results
ProcessImageResult Frame::processImageCall(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                           const int32_t* pixelIdxs, float tau, float beta, double alpha, bool wantInlierMap) {
    bind();
    ProcessImageResult r;
    const size_t P = (size_t)H_ * W_;
    std::vector<double> poses((size_t)objHyps * 6), gt = poseGT.getRodVecAndTrans();
    std::vector<uint8_t> ok(objHyps);
    r.imgIdx.assign(objHyps, {0, 0, 0, 0});
    r.sfScores.assign(objHyps, 0.0);
    if (wantInlierMap) r.inlierMap.assign(P, 0);
    double avg[6], ref[6], out4[4];
    int32_t sd = 0;
    check(dsac_process_images(ctx_, objHyps, seed, (float)inlierThreshold2D, 1 << 20, (float)CNN_OBJ_MAXINPUT, tau, beta, alpha, pixelIdxs, refSteps, inlierCount,
                              50, gt.data(), poses.data(), &r.imgIdx[0][0], ok.data(), nullptr, nullptr, r.sfScores.data(), &r.sfEntropy, avg, ref, &sd,
                              wantInlierMap ? r.inlierMap.data() : nullptr, out4),
          "dsac_process_images");
    r.hyps.resize(objHyps);
    for (int h = 0; h < objHyps; h++) r.hyps[h] = pose_at(&poses[(size_t)h * 6]);
    r.avgHyp = pose_at(avg);
    r.refAvgHyp = pose_at(ref);
    r.refStepsDone = sd;
    r.loss = out4[0]; r.rotErr = out4[1]; r.tErr = out4[2]; r.correct = out4[3] > 0.5;
    return r;
}

// ---- FrameBatch ----------------------------------------------------------------------------------------------------------------------
namespace {
// dsac_set_option("device_args", 1) for the calls of a scope whose pointer arguments all live in HBM; back to argument detection on leaving it -- the
// context is shared with Frame, whose calls take host arrays
struct DeviceArgsScope {
    Context& C;
    int before;  // what the context's user had set: restored on leaving the scope (round 4 reset it to 0 and so clobbered a caller's own setting)
    explicit DeviceArgsScope(Context& ctx) : C(ctx), before(ctx.option("device_args", 0)) { if (!before) C.setOption("device_args", 1); }
    ~DeviceArgsScope() { if (!before) { try { C.setOption("device_args", 0); } catch (...) {} } }
    DeviceArgsScope(const DeviceArgsScope&) = delete;
    DeviceArgsScope& operator=(const DeviceArgsScope&) = delete;
};
}  // namespace

FrameBatch::FrameBatch(Context& ctx, int frames, int H, int W, const Camera& cam, int objHyps, int refSteps, const std::vector<int32_t>& pixelIdxs,
                       int maxFramesPerCall, const FrameBatchOptions& opt)
    : C_(ctx), F_(frames), H_(H), W_(W), N_(objHyps), refSteps_(refSteps), maxCall_(maxFramesPerCall), cam_(cam), opt_(opt) {
    if (frames <= 0 || H <= 0 || W <= 0 || objHyps <= 0 || refSteps < 0 || maxFramesPerCall <= 0) throw Error(DSAC_ERR_INVALID, "FrameBatch: bad sizes");
    if (maxCall_ > F_) maxCall_ = F_;
    if (maxCall_ > 1 && objHyps % 128 != 0) throw Error(DSAC_ERR_INVALID, "FrameBatch: objHyps must be a multiple of 128 when more than one frame goes into a call");
    const size_t P = (size_t)H * W, F = (size_t)frames, N = (size_t)objHyps;
    if ((size_t)refSteps * P > pixelIdxs.size()) throw Error(DSAC_ERR_INVALID, "FrameBatch: pixelIdxs holds fewer than refSteps permutations");
    xyz_.resize(ctx, F * P * 3);
    if (opt.sampling) uv_.resize(ctx, F * P * 2);
    gt_.resize(ctx, F * 6);
    perm_.resize(ctx, (size_t)refSteps * P);
    perm_.upload(pixelIdxs.data(), (size_t)refSteps * P);
    poses_.resize(ctx, F * N * 6);
    sets_.resize(ctx, F * N * 4);
    ok_.resize(ctx, F * N);
    scores_.resize(ctx, F * N);
    w_.resize(ctx, F * N);
    entropy_.resize(ctx, F);
    avg_.resize(ctx, F * 6);
    ref_.resize(ctx, F * 6);
    out4_.resize(ctx, F * 4);
    stepsDone_.resize(ctx, F);
    if (opt.errorImages) err_.resize(ctx, (size_t)maxCall_ * N * P);  // one buffer: K2 of a call overwrites it in stream order (the tail never reads it)
    if (opt.inlierMaps) maps_.resize(ctx, F * P);
    done_.assign(F, 0);
    C_.synchronize();  // pixelIdxs may be a temporary
}

void FrameBatch::setFrame(int f, const float* estObj, const Hypothesis& poseGT, const float* sampling) {
    if (f < 0 || f >= F_ || !estObj) throw Error(DSAC_ERR_INVALID, "FrameBatch::setFrame: bad frame index");
    if (opt_.sampling != (sampling != nullptr)) throw Error(DSAC_ERR_INVALID, "FrameBatch::setFrame: a sampling table goes with FrameBatchOptions::sampling, and only with it");
    const size_t P = (size_t)H_ * W_;
    const std::vector<double> g = poseGT.getRodVecAndTrans();
    // a deferred tail of the previous processImages may still refine against this buffer: dsac_copy_async orders every copy behind a tail in flight (since
    // round 5, whichever way the copy goes), so the uploads below are safe right after a deferred call (tests/test_gpu_host_shim.py)
    xyz_.upload(estObj, P * 3, (size_t)f * P * 3);
    if (sampling) uv_.upload(sampling, P * 2, (size_t)f * P * 2);
    gt_.upload(g.data(), 6, (size_t)f * 6);
    C_.synchronize();  // pageable sources: stable once this returns
    done_[f] = 0;
}

void FrameBatch::processImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, float tau, float beta, double alpha) {
    if (first < 0 || count <= 0 || first + count > F_ || count > maxCall_) throw Error(DSAC_ERR_INVALID, "FrameBatch::processImages: bad frame range");
    const size_t P = (size_t)H_ * W_, N = (size_t)N_, f0 = (size_t)first;
    dsac_ctx* c = C_.get();
    C_.setOption("pi_defer_tail", opt_.deferTail ? (opt_.deferScoreTail ? 2 : 1) : 0);
    // with the score tail deferred every result row of the previous call is still being written: the same frames again (one call per pass) go in order
    if (opt_.deferTail && opt_.deferScoreTail && first < lastFirst_ + lastCount_ && lastFirst_ < first + count) C_.check(dsac_join_tail(c), "dsac_join_tail");
    lastFirst_ = first; lastCount_ = count;
    unsigned flags = DSAC_FRAME_BORROW;
    if (opt_.quantiseInt16) flags |= DSAC_FRAME_QUANTISE_INT16;  // in place; idempotent
    C_.check(dsac_set_frames(c, count, xyz_.data() + f0 * P * 3, opt_.sampling ? uv_.data() + f0 * P * 2 : nullptr, opt_.sampling ? 1 : 0, H_, W_, cam_.fx, cam_.fy, cam_.cx,
                             cam_.cy, flags), "dsac_set_frames");
    C_.setBound(this);
    const DeviceArgsScope devArgs(C_);  // every argument below lives in HBM: no pointer query per argument
    const int rcP = dsac_process_images(c, N_, seedOfFrame0 + (uint64_t)first, (float)inlierThreshold2D, 1 << 20, (float)CNN_OBJ_MAXINPUT, tau, beta, alpha, perm_.data(),
                                 refSteps_, inlierCount, 50, gt_.data() + f0 * 6, poses_.data() + f0 * N * 6, sets_.data() + f0 * N * 4, ok_.data() + f0 * N,
                                 opt_.errorImages ? err_.data() : nullptr, scores_.data() + f0 * N, w_.data() + f0 * N, entropy_.data() + f0, avg_.data() + f0 * 6,
                                 ref_.data() + f0 * 6, stepsDone_.data() + f0, opt_.inlierMaps ? maps_.data() + f0 * P : nullptr, out4_.data() + f0 * 4);
    C_.check(rcP, "dsac_process_images");
    for (int f = first; f < first + count; f++) done_[f] = 1;
}

void FrameBatch::backward(int first, int count, int inlierThreshold2D, int inlierCount, float subSampleFactor, float tau, float beta, double alpha) {
    if (first < 0 || count <= 0 || first + count > F_ || count > maxCall_) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: bad frame range");
    if (!opt_.inlierMaps) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: the batch was made without FrameBatchOptions::inlierMaps");
    for (int f = first; f < first + count; f++)
        if (!done_[f]) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: processImages has not run on a frame of the range");
    const size_t P = (size_t)H_ * W_, N = (size_t)N_, f0 = (size_t)first;
    dsac_ctx* c = C_.get();
    ensureBackwardBuffers();
    unsigned flags = DSAC_FRAME_BORROW;
    if (opt_.quantiseInt16) flags |= DSAC_FRAME_QUANTISE_INT16;
    // any entry point but dsac_process_images orders the stream behind a deferred tail: the refined poses and inlier maps are complete for what follows
    C_.check(dsac_set_frames(c, count, xyz_.data() + f0 * P * 3, opt_.sampling ? uv_.data() + f0 * P * 2 : nullptr, opt_.sampling ? 1 : 0, H_, W_, cam_.fx, cam_.fy, cam_.cx,
                             cam_.cy, flags), "dsac_set_frames");
    C_.setBound(this);
    double* grad = grad_.data() + f0 * P * 3;
    const DeviceArgsScope devArgs(C_);
    C_.check(dsac_fill_zero_async(c, grad, (size_t)count * P * 3 * sizeof(double)), "dsac_fill_zero_async");
    const int n = count * N_;
    // path I and the softmax backward (train_ransac_softam.cpp:294-376); g comes back scaled by alpha: the score is alpha x the soft-inlier count
    C_.check(dsac_backward_path1(c, n, poses_.data() + f0 * N * 6, sets_.data() + f0 * N * 4, w_.data() + f0 * N, avg_.data() + f0 * 6, ref_.data() + f0 * 6,
                                 gt_.data() + f0 * 6, perm_.data(), refSteps_, inlierCount, 50, (float)inlierThreshold2D, maps_.data() + f0 * P, subSampleFactor, 0.001f,
                                 2.f, alpha, dpnp_.data(), grad, g_.data(), nullptr, nullptr),
             "dsac_backward_path1");
    // path II: score gradients -> error images -> scene coordinates (dScore, :379-383), the error-image gradient formed in-kernel
    C_.check(dsac_soft_score_backward(c, n, poses_.data() + f0 * N * 6, sets_.data() + f0 * N * 4, g_.data(), (float)CNN_OBJ_MAXINPUT, tau, beta, dpnp_.data(), 0u, grad),
             "dsac_soft_score_backward");
    lastFirst_ = 0; lastCount_ = 0;  // the stream is ordered behind every tail now
}

// ---- the score-CNN seam on the batch ----------------------------------------------------------------------------------------------------
void FrameBatch::bindRange(int first, int count) {
    const size_t P = (size_t)H_ * W_, f0 = (size_t)first;
    unsigned flags = DSAC_FRAME_BORROW;
    if (opt_.quantiseInt16) flags |= DSAC_FRAME_QUANTISE_INT16;  // in place; idempotent
    C_.check(dsac_set_frames(C_.get(), count, xyz_.data() + f0 * P * 3, opt_.sampling ? uv_.data() + f0 * P * 2 : nullptr, opt_.sampling ? 1 : 0, H_, W_, cam_.fx, cam_.fy,
                             cam_.cx, cam_.cy, flags), "dsac_set_frames");
    C_.setBound(this);
}

void FrameBatch::scoreImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, float tau, float beta) {
    if (first < 0 || count <= 0 || first + count > F_ || count > maxCall_) throw Error(DSAC_ERR_INVALID, "FrameBatch::scoreImages: bad frame range");
    if (!opt_.errorImages) throw Error(DSAC_ERR_INVALID, "FrameBatch::scoreImages: the batch was made without FrameBatchOptions::errorImages");
    const size_t N = (size_t)N_, f0 = (size_t)first;
    dsac_ctx* c = C_.get();
    C_.setOption("pi_defer_tail", opt_.deferTail ? (opt_.deferScoreTail ? 2 : 1) : 0);
    if (opt_.deferTail && opt_.deferScoreTail && first < lastFirst_ + lastCount_ && lastFirst_ < first + count) C_.check(dsac_join_tail(c), "dsac_join_tail");
    lastFirst_ = first; lastCount_ = count;
    // one slice per FRAME (like scores_ / w_), not one buffer per call: with the score tail deferred (pi_defer_tail = 2) K3 of this call reads its scores on
    // the tail stream while the NEXT call's begin already writes its sums -- consecutive calls on different ranges must not share the array
    if (soft_.size() == 0) soft_.resize(C_, (size_t)F_ * N);
    bindRange(first, count);
    const DeviceArgsScope devArgs(C_);
    C_.check(dsac_process_images_begin(c, N_, seedOfFrame0 + (uint64_t)first, (float)inlierThreshold2D, 1 << 20, (float)CNN_OBJ_MAXINPUT, tau, beta, poses_.data() + f0 * N * 6,
                                       sets_.data() + f0 * N * 4, ok_.data() + f0 * N, err_.data(), soft_.data() + f0 * N),
             "dsac_process_images_begin");
    seamFirst_ = first; seamCount_ = count;
}

void FrameBatch::finishImages(int first, int count, const double* scoresDevice, int inlierThreshold2D, int inlierCount, double scale) {
    if (first != seamFirst_ || count != seamCount_) throw Error(DSAC_ERR_INVALID, "FrameBatch::finishImages: the range is not the one scoreImages ran on");
    if (!scoresDevice) throw Error(DSAC_ERR_INVALID, "FrameBatch::finishImages: scores is NULL");
    const size_t P = (size_t)H_ * W_, N = (size_t)N_, f0 = (size_t)first;
    dsac_ctx* c = C_.get();
    const DeviceArgsScope devArgs(C_);
    C_.check(dsac_process_images_finish(c, N_, scoresDevice, scale, perm_.data(), refSteps_, inlierCount, 50, (float)inlierThreshold2D, gt_.data() + f0 * 6,
                                        poses_.data() + f0 * N * 6, w_.data() + f0 * N, entropy_.data() + f0, avg_.data() + f0 * 6, ref_.data() + f0 * 6,
                                        stepsDone_.data() + f0, opt_.inlierMaps ? maps_.data() + f0 * P : nullptr, out4_.data() + f0 * 4),
             "dsac_process_images_finish");
    for (int f = first; f < first + count; f++) done_[f] = 1;
}

void FrameBatch::processImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, const ScoreModel& model, float tau, float beta) {
    if (!model.forward) throw Error(DSAC_ERR_INVALID, "FrameBatch::processImages: the score model has no forward function");
    scoreImages(first, count, seedOfFrame0, inlierThreshold2D, tau, beta);  // (tau, beta): the soft-inlier sums scoreImages leaves on the side (softInlierModel reads them)
    const double* scores = model.forward(err_.data(), count * N_, H_, W_);  // core/cnn_softam.h:1072
    finishImages(first, count, scores, inlierThreshold2D, inlierCount, model.scale);
}

ScoreModel FrameBatch::softInlierModel(float tau, float beta, double alpha) {
    ScoreModel m;
    m.scale = alpha;
    // scoreImages evaluated the sums with its own (tau, beta): the model's must be the same pair (the defaults are)
    m.forward = [this](const float*, int, int, int) { return soft_.data() + (size_t)seamFirst_ * N_; };
    m.backward = [this, tau, beta](const double* g, const float* err, int nMaps, int, int) -> const float* {
        const size_t P = (size_t)H_ * W_;
        if (dErr_.size() == 0) dErr_.resize(C_, (size_t)maxCall_ * N_ * P);
        const DeviceArgsScope devArgs(C_);
        C_.check(dsac_soft_score_derr(C_.get(), nMaps, g, err, (float)CNN_OBJ_MAXINPUT, tau, beta, dErr_.data()), "dsac_soft_score_derr");
        return dErr_.data();
    };
    return m;
}

void FrameBatch::ensureBackwardBuffers() {
    if (grad_.size() != 0) return;
    const size_t P = (size_t)H_ * W_, N = (size_t)N_;
    grad_.resize(C_, (size_t)F_ * P * 3);
    dpnp_.resize(C_, (size_t)maxCall_ * N * 72);
    g_.resize(C_, (size_t)maxCall_ * N);
}

void FrameBatch::backward(int first, int count, int inlierThreshold2D, int inlierCount, float subSampleFactor, const ScoreModel& model, bool referenceIndexQuirk) {
    if (first < 0 || count <= 0 || first + count > F_ || count > maxCall_) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: bad frame range");
    if (!opt_.inlierMaps) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: the batch was made without FrameBatchOptions::inlierMaps");
    if (!model.backward) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: the score model has no backward function");
    if (first != seamFirst_ || count != seamCount_) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: the error images in HBM belong to another frame range");
    for (int f = first; f < first + count; f++)
        if (!done_[f]) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: processImages has not run on a frame of the range");
    const size_t P = (size_t)H_ * W_, N = (size_t)N_, f0 = (size_t)first;
    dsac_ctx* c = C_.get();
    ensureBackwardBuffers();
    bindRange(first, count);  // any entry point but dsac_process_images orders the stream behind a deferred tail
    double* grad = grad_.data() + f0 * P * 3;
    const int n = count * N_;
    {
        const DeviceArgsScope devArgs(C_);
        C_.check(dsac_fill_zero_async(c, grad, (size_t)count * P * 3 * sizeof(double)), "dsac_fill_zero_async");
        // path I and the softmax backward (train_ransac_softam.cpp:294-376): g = dLoss / d(scale * score); the model's scores enter scaled
        C_.check(dsac_backward_path1(c, n, poses_.data() + f0 * N * 6, sets_.data() + f0 * N * 4, w_.data() + f0 * N, avg_.data() + f0 * 6, ref_.data() + f0 * 6,
                                     gt_.data() + f0 * 6, perm_.data(), refSteps_, inlierCount, 50, (float)inlierThreshold2D, maps_.data() + f0 * P, subSampleFactor, 0.001f,
                                     2.f, model.scale, dpnp_.data(), grad, g_.data(), nullptr, nullptr),
                 "dsac_backward_path1");
    }
    const float* dErr = model.backward(g_.data(), err_.data(), n, H_, W_);  // train_ransac_softam.cpp:378-381
    if (!dErr) throw Error(DSAC_ERR_INVALID, "FrameBatch::backward: the score model returned no gradient images");
    {
        const DeviceArgsScope devArgs(C_);
        C_.check(dsac_score_backward(c, n, poses_.data() + f0 * N * 6, sets_.data() + f0 * N * 4, dErr, dpnp_.data(), referenceIndexQuirk ? DSAC_BWD_QUIRK_TRANSPOSE : 0u, grad),
                 "dsac_score_backward");  // :382-383
    }
    lastFirst_ = 0; lastCount_ = 0;
}

std::vector<double> FrameBatch::gradients(int f) {
    if (f < 0 || f >= F_ || grad_.size() == 0) throw Error(DSAC_ERR_INVALID, "FrameBatch::gradients: no backward pass has run");
    const size_t n = (size_t)H_ * W_ * 3;
    return grad_.toHost(n, (size_t)f * n);
}

void FrameBatch::copyFrameFrom(const FrameBatch& src, int srcFrame, int dstFrame) {
    if (src.H_ != H_ || src.W_ != W_ || src.opt_.sampling != opt_.sampling || srcFrame < 0 || srcFrame >= src.F_ || dstFrame < 0 || dstFrame >= F_)
        throw Error(DSAC_ERR_INVALID, "FrameBatch::copyFrameFrom: geometry or frame index does not match");
    const size_t n = (size_t)H_ * W_ * 3;
    // dsac_copy_async orders the stream behind a deferred tail first (the previous step's refinement may still read the destination)
    C_.copy(xyz_.data() + (size_t)dstFrame * n, src.xyz_.data() + (size_t)srcFrame * n, n * sizeof(float));
    C_.copy(gt_.data() + (size_t)dstFrame * 6, src.gt_.data() + (size_t)srcFrame * 6, 6 * sizeof(double));
    if (opt_.sampling) C_.copy(uv_.data() + (size_t)dstFrame * n / 3 * 2, src.uv_.data() + (size_t)srcFrame * n / 3 * 2, n / 3 * 2 * sizeof(float));
    done_[dstFrame] = 0;
}

void FrameBatch::gatherFramesFrom(const FrameBatch& src, const std::vector<int32_t>& srcFrames, int dstFirst) {
    const int n = (int)srcFrames.size();
    if (src.H_ != H_ || src.W_ != W_ || src.opt_.sampling != opt_.sampling || dstFirst < 0 || dstFirst + n > F_)
        throw Error(DSAC_ERR_INVALID, "FrameBatch::gatherFramesFrom: geometry or frame range does not match");
    for (int32_t f : srcFrames)
        if (f < 0 || f >= src.F_) throw Error(DSAC_ERR_INVALID, "FrameBatch::gatherFramesFrom: bad source frame");
    if (n == 0) return;
    const size_t P = (size_t)H_ * W_, d0 = (size_t)dstFirst;
    dsac_ctx* c = C_.get();
    C_.check(dsac_gather_rows(c, xyz_.data() + d0 * P * 3, src.xyz_.data(), P * 3 * sizeof(float), n, srcFrames.data()), "dsac_gather_rows");
    C_.check(dsac_gather_rows(c, gt_.data() + d0 * 6, src.gt_.data(), 6 * sizeof(double), n, srcFrames.data()), "dsac_gather_rows");
    if (opt_.sampling) C_.check(dsac_gather_rows(c, uv_.data() + d0 * P * 2, src.uv_.data(), P * 2 * sizeof(float), n, srcFrames.data()), "dsac_gather_rows");
    for (int k = 0; k < n; k++) done_[dstFirst + k] = 0;
}

void FrameBatch::processAll(uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, float tau, float beta, double alpha) {
    for (int f = 0; f < F_; f += maxCall_) processImages(f, std::min(maxCall_, F_ - f), seedOfFrame0, inlierThreshold2D, inlierCount, tau, beta, alpha);
}

void FrameBatch::processAll(uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, const ScoreModel& model, float tau, float beta) {
    for (int f = 0; f < F_; f += maxCall_) processImages(f, std::min(maxCall_, F_ - f), seedOfFrame0, inlierThreshold2D, inlierCount, model, tau, beta);
}

void FrameBatch::synchronize() {
    C_.check(dsac_join_tail(C_.get()), "dsac_join_tail");
    C_.synchronize();
}

std::vector<ProcessImageResult> FrameBatch::results(bool perHypothesis) {
    C_.check(dsac_join_tail(C_.get()), "dsac_join_tail");
    const size_t F = (size_t)F_, N = (size_t)N_, P = (size_t)H_ * W_;
    const std::vector<double> ent = entropy_.toHost(F), avg = avg_.toHost(F * 6), ref = ref_.toHost(F * 6), o4 = out4_.toHost(F * 4);
    const std::vector<int32_t> sd = stepsDone_.toHost(F);
    std::vector<double> poses, w;
    std::vector<int32_t> sets, maps;
    if (perHypothesis) { poses = poses_.toHost(F * N * 6); w = w_.toHost(F * N); sets = sets_.toHost(F * N * 4); }
    if (opt_.inlierMaps) maps = maps_.toHost(F * P);
    std::vector<ProcessImageResult> out(F);
    for (size_t f = 0; f < F; f++) {
        ProcessImageResult& r = out[f];
        if (!done_[f]) continue;
        r.sfEntropy = ent[f];
        r.avgHyp = pose_at(&avg[f * 6]);
        r.refAvgHyp = pose_at(&ref[f * 6]);
        r.refStepsDone = sd[f];
        r.loss = o4[f * 4]; r.rotErr = o4[f * 4 + 1]; r.tErr = o4[f * 4 + 2]; r.correct = o4[f * 4 + 3] > 0.5;
        if (perHypothesis) {
            r.hyps.resize(N);
            r.imgIdx.resize(N);
            for (size_t h = 0; h < N; h++) {
                r.hyps[h] = pose_at(&poses[(f * N + h) * 6]);
                for (int k = 0; k < 4; k++) r.imgIdx[h][k] = sets[(f * N + h) * 4 + k];
            }
            r.sfScores.assign(w.begin() + f * N, w.begin() + (f + 1) * N);
        }
        if (opt_.inlierMaps) r.inlierMap.assign(maps.begin() + f * P, maps.begin() + (f + 1) * P);
    }
    return out;
}

std::vector<double> Frame::backward(const ProcessImageResult& fwd, const Hypothesis& poseGT, int inlierThreshold2D, int inlierCount, int refSteps,
                                    float subSampleFactor, const std::vector<int32_t>& pixelIdxs, float tau, float beta, double alpha,
                                    bool referenceIndexQuirk) {
    bind();
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

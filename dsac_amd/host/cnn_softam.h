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
//   processImage          cnn_softam.h:960                       Frame::processImage (one image, one ABI call)
//   the loop over images  test_ransac_softam.cpp:97-157          FrameBatch::processImages (16 images per launch chain, HBM-resident)
//
// One engine context serves the whole program (Context::shared): dsac_create -- device query, stream, scratch allocation -- is paid once,
// a Frame only uploads its coordinate map and binds it (DSAC_FRAME_BORROW) before a call.
//
// Every call throws dsac::Error (carrying the dsac_status and dsac_last_error text) on failure; nothing is
// computed on the CPU here -- without a gfx950 device the Frame constructor throws.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
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

// One engine context for the life of the program (or of whoever owns it): the device query, the stream and the library's scratch buffers
// are created once; Frames and FrameBatches bind their HBM-resident coordinate maps to it call by call.  Single-threaded, like dsac_ctx.
class Context {
public:
    explicit Context(int device = 0);  // throws dsac::Error without a gfx950 device (no CPU fallback)
    ~Context();
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    static Context& shared(int device = 0);  // the process-wide context of a device, created on first use, destroyed at exit

    dsac_ctx* get() { return ctx_; }
    void check(int rc, const char* what);
    void synchronize();
    // ThreadRand::forceInit(seed) with `threads` OpenMP threads (core/thread_rand.cpp:33-57): the reference's generators, kept in the context and run on through
    // successive images by Frame::sampleHypothesesRefStream; discardRand = draws reference code outside the sampling loop took from generator `thread`
    void forceInitRand(unsigned seed = 1305, int threads = 1) { check(dsac_refstream_init(ctx_, seed, threads), "dsac_refstream_init"); }
    void discardRand(int thread, unsigned long long outputs32) { check(dsac_refstream_discard(ctx_, thread, outputs32), "dsac_refstream_discard"); }
    void setOption(const char* key, int value);
    int option(const char* key, int unset = 0) const;  // the value last given to setOption (the C ABI has no getter; scopes that change a knob restore it from here)
    // HBM / page-locked host buffers and stream-ordered copies (dsac_device_alloc, dsac_host_alloc, dsac_copy_async)
    void* deviceAlloc(size_t bytes);
    void deviceFree(void* p) noexcept;
    void* hostAlloc(size_t bytes);
    void hostFree(void* p) noexcept;
    void copy(void* dst, const void* src, size_t bytes);
    void zero(void* dst, size_t bytes);
    // which Frame / FrameBatch owns the frame currently set in the context (so that re-binding is skipped when nothing changed)
    const void* boundTo() const { return bound_; }
    void setBound(const void* owner) { bound_ = owner; }

private:
    dsac_ctx* ctx_ = nullptr;
    const void* bound_ = nullptr;
    std::map<std::string, int> options_;
};

// A typed buffer in HBM, freed with its owner.  upload / download are ordered on the context's stream; download() waits for the data.
template <typename T>
class DeviceArray {
public:
    DeviceArray() = default;
    DeviceArray(Context& c, size_t n) { resize(c, n); }
    ~DeviceArray() { release(); }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    DeviceArray(DeviceArray&& o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceArray& operator=(DeviceArray&& o) noexcept { if (this != &o) { release(); c_ = o.c_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    void resize(Context& c, size_t n) {
        if (n == n_ && c_ == &c) return;
        release();
        c_ = &c;
        p_ = n ? static_cast<T*>(c.deviceAlloc(n * sizeof(T))) : nullptr;
        n_ = n;
    }
    void release() noexcept { if (p_ && c_) c_->deviceFree(p_); p_ = nullptr; n_ = 0; }
    T* data() const { return p_; }
    size_t size() const { return n_; }
    void upload(const T* src, size_t count, size_t at = 0) { c_->copy(p_ + at, src, count * sizeof(T)); }
    void download(T* dst, size_t count, size_t at = 0) const { c_->copy(dst, p_ + at, count * sizeof(T)); c_->synchronize(); }
    std::vector<T> toHost(size_t count, size_t at = 0) const { std::vector<T> v(count); if (count) download(v.data(), count, at); return v; }

private:
    Context* c_ = nullptr;
    T* p_ = nullptr;
    size_t n_ = 0;
};

class Frame {
public:
    // estObj: H*W*3 float32 mm; sampling: H*W*2 float32 (u,v) or nullptr for the full-resolution grid.  The map is uploaded to HBM once;
    // the context is the process-wide one of `device` (Context::shared) or the caller's.
    Frame(const float* estObj, const float* sampling, int H, int W, const Camera& cam, int device = 0, bool quantiseInt16 = false);
    Frame(Context& ctx, const float* estObj, const float* sampling, int H, int W, const Camera& cam, bool quantiseInt16 = false);
    ~Frame();
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;

    int rows() const { return H_; }
    int cols() const { return W_; }

    // sampling loop of processImage (cnn_softam.h:1010-1060); returns the per-hypothesis success flags
    std::vector<uint8_t> sampleHypotheses(int objHyps, uint64_t seed, int inlierThreshold2D, std::vector<cv_trans_t>& hyps,
                                          std::vector<std::array<int32_t, 4>>& imgIdx, int maxTries = 1 << 20);
    // the same loop drawing from the REFERENCE'S OWN generators (ThreadRand, core/thread_rand.cpp:40-69: std::mt19937(seed + t) per OpenMP thread; set up with
    // forceInitRand below): minimal sets bit-identical to the reference's for the same seed and thread count, no replay file needed.  subSampleOutputs: what
    // thread 0 drew BEFORE the loop in the reference's processImage -- stochasticSubSample's two drand per cell = 6400 32-bit outputs for its 40 x 40 grid
    // (core/cnn_softam.h:283-309, :1003); 0 when the sampling grid was not drawn from the stream
    std::vector<uint8_t> sampleHypothesesRefStream(int objHyps, int inlierThreshold2D, std::vector<cv_trans_t>& hyps, std::vector<std::array<int32_t, 4>>& imgIdx,
                                                   unsigned long long subSampleOutputs = 0, long long maxAttempts = 1ll << 24);
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
    // the same with the refinement permutations already in HBM (refSteps * H*W indices): an evaluation loop uploads them once, not per image
    ProcessImageResult processImage(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                    const DeviceArray<int32_t>& pixelIdxsDevice, float tau = 10.f, float beta = 0.5f, double alpha = 0.1, bool wantInlierMap = true);
    // The backward section of the trainer, core/train_ransac_softam.cpp:288-394: dLoss/d(scene coordinates), H*W x 3, from the
    // forward pass's results.  Path I: dLossMax . (dRefineObj + dRefineHyp . sum_h w_h dPNP_h); path II: softmax backward -> score
    // gradients -> (soft-inlier) score backward.  referenceIndexQuirk reproduces the transposed pixel index of path II (:628,641).
    std::vector<double> backward(const ProcessImageResult& fwd, const Hypothesis& poseGT, int inlierThreshold2D, int inlierCount, int refSteps,
                                 float subSampleFactor, const std::vector<int32_t>& pixelIdxs, float tau = 10.f, float beta = 0.5f,
                                 double alpha = 0.1, bool referenceIndexQuirk = false);

    // the N error images in HBM (N*H*W floats, hypothesis-major): what the score CNN consumes in place (lua_calls.h:98-104) -- no PCIe round trip
    void getDiffMapsDevice(const std::vector<cv_trans_t>& hyps, DeviceArray<float>& out);

    dsac_ctx* context() { bind(); return ctx_; }
    Context& engine() { return *C_; }

protected:
    ProcessImageResult processImageCall(const Hypothesis& poseGT, int objHyps, uint64_t seed, int inlierThreshold2D, int inlierCount, int refSteps,
                                        const int32_t* pixelIdxsHostOrDevice, float tau, float beta, double alpha, bool wantInlierMap);
    void check(int rc, const char* what);
    void bind();  // make this frame the context's current one (a borrowed-pointer rebind; nothing is copied)
    Context* C_ = nullptr;
    dsac_ctx* ctx_ = nullptr;
    int H_, W_;
    Camera cam_;
    DeviceArray<float> xyz_, uv_;
    bool quantise_ = false, quantised_ = false;
};

// The loop over images of core/test_ransac_softam.cpp:97-157 as launches over image BATCHES: F coordinate maps of one geometry live in HBM,
// processImages(first, count, ...) enqueues processImage (core/cnn_softam.h:960-1179) for `count` of them as ONE chain of launches
// (dsac_set_frames + dsac_process_images: K1, K2, K3, K6 with a wave per image, K7), every output stays in HBM, the refinement tail of a
// call runs under sampling and scoring of the next one (pi_defer_tail), and nothing blocks the host until results().  Image i draws from
// the random stream of seed + i, so the results equal Frame::processImage(seed + i) image by image, bit for bit.
struct FrameBatchOptions {
    bool errorImages = true;   // write the N error images per frame (the score CNN's input; one buffer shared by all calls)
    bool inlierMaps = false;   // keep the refinement's inlier maps (needed by the training backward only)
    bool deferTail = true;     // dsac_set_option("pi_defer_tail", ...): the refinement tail of a call runs under K1 / K2 of the next one
    bool deferScoreTail = true;  // ... = 2: the score reduction and K3 as well (consecutive calls write different frames' result rows); 64 images in
                                 // batches of 16: 65.6 (in order) / 62.8 / 62.6 us per image, in batches of 4: 87.9 / - / 65.7 (profiles/r04_host_driver_defer_ab.txt)
    bool quantiseInt16 = false;
    bool sampling = false;     // every frame brings its own H*W x 2 table of image positions (the reference's sub-sampled 40x40 maps: stochasticSubSample,
                               // core/cnn_softam.h:283-309, draws them per image); false: the full-resolution grid, cell (x, y) at pixel (x, y)
};

// The score model at the seam of the batched path -- the reference's score CNN (core/cnn_softam.h:1072: forward(diffMaps); core/train_ransac_softam.cpp:
// 378-383: backward -> dScore), or anything else that turns error images into scores.  Both functions are called with DEVICE pointers and must enqueue their
// work on the engine's stream (dsac_get_stream) or order themselves against it; the pointers they return must stay valid until the batch's next call --
// and with FrameBatchOptions::deferScoreTail (the default) the SCORES of call k are read by K3 on the tail stream while call k + 1 is already running: a
// model must then hand out a buffer that it does not rewrite before the call AFTER the next (two alternating buffers, or one slice per frame as
// softInlierModel does), or the batch must be made with deferScoreTail = false.
//   forward : nMaps x H*W float32 error images (hypothesis-major, each map row-major: the order of core/lua_calls.h:98-104) -> nMaps scores (double)
//   backward: nMaps score gradients (double) and the same error images -> nMaps x H*W float32 gradient images, (n, y, x) order
struct ScoreModel {
    std::function<const double*(const float* errDevice, int nMaps, int H, int W)> forward;
    std::function<const float*(const double* scoreGradientsDevice, const float* errDevice, int nMaps, int H, int W)> backward;
    double scale = 1.0;  // softMax(scale * scores)
};

class FrameBatch {
public:
    FrameBatch(Context& ctx, int frames, int H, int W, const Camera& cam, int objHyps, int refSteps, const std::vector<int32_t>& pixelIdxs,
               int maxFramesPerCall = 16, const FrameBatchOptions& opt = FrameBatchOptions());
    int frames() const { return F_; }
    // stream-ordered upload of frame f's coordinate map (H*W*3 floats, mm) and ground truth
    void setFrame(int f, const float* estObj, const Hypothesis& poseGT, const float* sampling = nullptr);  // sampling: FrameBatchOptions::sampling
    // enqueue processImage for frames [first, first + count), count <= maxFramesPerCall; returns immediately
    void processImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, float tau = 10.f, float beta = 0.5f,
                       double alpha = 0.1);
    // ---- the same with the score taken from OUTSIDE the library (the score-CNN seam; needs FrameBatchOptions::errorImages) ----
    // scoreImages: K1 + K2 of the range -> error images in HBM (errorImagesDevice) and, on the side, the soft-inlier sums (softInlierSumsDevice);
    // finishImages: K3 on scale * scores, the refinement and the loss.  processImages(.., model) = scoreImages -> model.forward -> finishImages.
    void scoreImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, float tau = 10.f, float beta = 0.5f);
    void finishImages(int first, int count, const double* scoresDevice, int inlierThreshold2D, int inlierCount, double scale = 1.0);
    void processImages(int first, int count, uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, const ScoreModel& model, float tau = 10.f, float beta = 0.5f);
    const double* softInlierSumsDevice() const { return soft_.data() + (size_t)(seamFirst_ < 0 ? 0 : seamFirst_) * N_; }  // of the most recent scoreImages, count*objHyps
    // the soft-inlier score dressed as an external model (forward: the sums scoreImages left; backward: dsac_soft_score_derr) -- the stand-in a host without
    // device code uses to drive the seam end to end; results equal the built-in score's to fp32 rounding
    ScoreModel softInlierModel(float tau = 10.f, float beta = 0.5f, double alpha = 0.1);
    // convenience: all frames in calls of maxFramesPerCall
    void processAll(uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, float tau = 10.f, float beta = 0.5f, double alpha = 0.1);
    void processAll(uint64_t seedOfFrame0, int inlierThreshold2D, int inlierCount, const ScoreModel& model, float tau = 10.f, float beta = 0.5f);  // through the seam
    void synchronize();  // joins the deferred tail and waits
    // waits and copies the results back; perHypothesis = false leaves hyps / imgIdx / sfScores empty (the evaluation program needs none of them)
    std::vector<ProcessImageResult> results(bool perHypothesis = true);
    const float* errorImagesDevice() const { return err_.data(); }  // of the most recent call, count*objHyps x H*W

    // ---- training (core/train_ransac_softam.cpp:288-394 for every frame of the range, one launch chain; needs FrameBatchOptions::inlierMaps) ----
    // Enqueue the backward pass of frames [first, first + count) behind their processImages: dLossMax at each refined pose, dRefineObj / dRefineHyp
    // (12 + 6n finite-difference replicas per frame), dPNP of the count * objHyps minimal sets, the path-I assembly, the softmax backward and the
    // soft-inlier score backward (K4) -- dsac_backward_path1 + dsac_soft_score_backward on the frame batch.  The gradient of the loss with respect to
    // frame f's scene coordinates (H*W x 3 doubles, what the reference hands to the scene-coordinate CNN's backward, :412) stays in HBM
    // (gradientsDevice); returns immediately.  count * objHyps hypotheses: objHyps a multiple of 16 and at most 256 when count > 1.
    void backward(int first, int count, int inlierThreshold2D, int inlierCount, float subSampleFactor, float tau = 10.f, float beta = 0.5f, double alpha = 0.1);
    // ... with the score model's own backward between the softmax backward and dScore (train_ransac_softam.cpp:378-383): path I -> score gradients ->
    // model.backward -> gradient images -> dsac_score_backward on the batch.  The range must be the one of the most recent processImages(.., model) /
    // scoreImages (its error images are what the model differentiates).  referenceIndexQuirk: dScore's transposed columns (cnn_softam.h:628,641); the
    // model then has to hand the images over transposed as the Lua bridge does (lua_calls.h:329-335).
    void backward(int first, int count, int inlierThreshold2D, int inlierCount, float subSampleFactor, const ScoreModel& model, bool referenceIndexQuirk = false);
    const double* scoreGradientsDevice() const { return g_.data(); }  // of the most recent backward, count*objHyps
    const double* gradientsDevice(int f) const { return grad_.data() + (size_t)f * H_ * W_ * 3; }
    std::vector<double> gradients(int f);  // waits, copies frame f's gradient back
    // device-to-device: frame `srcFrame` of another batch of the same geometry becomes frame `dstFrame` of this one (coordinate map + ground truth),
    // in stream order -- how a training step draws its frames from a data set that stays resident in HBM
    void copyFrameFrom(const FrameBatch& src, int srcFrame, int dstFrame);
    // the same for frames dstFirst, dstFirst + 1, ... at once: dsac_gather_rows, one launch per array instead of one copy per frame and array
    void gatherFramesFrom(const FrameBatch& src, const std::vector<int32_t>& srcFrames, int dstFirst = 0);

private:
    Context& C_;
    int F_, H_, W_, N_, refSteps_, maxCall_;
    int lastFirst_ = 0, lastCount_ = 0;  // the frame range of the previous call (its result rows are still being written with a deferred score tail)
    Camera cam_;
    FrameBatchOptions opt_;
    DeviceArray<float> xyz_, uv_, err_;
    DeviceArray<int32_t> perm_, sets_, stepsDone_, maps_;
    DeviceArray<uint8_t> ok_;
    DeviceArray<double> gt_, poses_, scores_, w_, entropy_, avg_, ref_, out4_;
    DeviceArray<double> grad_, dpnp_, g_;  // training: F x H*W x 3 gradient, maxCall x objHyps x 72 dPNP, maxCall x objHyps score gradients
    DeviceArray<double> soft_;             // the seam: soft-inlier sums, one slice per frame (frames x objHyps): a deferred score tail reads call k's while call k + 1 writes its own
    DeviceArray<float> dErr_;              // softInlierModel's gradient images (maxCall x objHyps x H*W), allocated on first use
    int seamFirst_ = -1, seamCount_ = 0;   // the range whose error images err_ holds
    void bindRange(int first, int count);
    void ensureBackwardBuffers();
    std::vector<uint8_t> done_;
};

// host-side forms of the two trivial reference functions (cnn_softam.h:535-553, :80-88)
std::vector<double> softMax(const std::vector<double>& scores);
double entropy(const std::vector<double>& dist);
// the reference's permutation stream: one default-seeded std::mt19937, Fisher-Yates per step (cnn_softam.h:1104-1114)
std::vector<int32_t> refinePermutations(int P, int refSteps);

}  // namespace dsac

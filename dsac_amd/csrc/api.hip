// api.hip -- the C ABI of libdsac_hip.so (include/dsac_hip.h): context, device scratch, host/device
// pointer handling, and the launch sequences.  No CPU compute path exists in this library.
#include "../../include/dsac_hip.h"
#include "kernels.h"
#include "refstream.h"

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <deque>
#include <string>
#include <vector>
// std::uniform_int_distribution of the libstdc++ this library is built against: GCC >= 11 draws by Lemire's method, older ones by scaling and division
// (refstream.h); "refstream_mode" overrides
#if defined(_GLIBCXX_RELEASE) && _GLIBCXX_RELEASE < 11
#define DSAC_RS_DEFAULT_MODE 1
#else
#define DSAC_RS_DEFAULT_MODE 0
#endif

namespace {

thread_local std::string g_last_error;

// A growable device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

}  // namespace

struct dsac_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    hipDeviceProp_t prop;

    // frame
    bool have_frame = false;
    dk::FrameDev F{};
    DevBuf frame_xyz, frame_uv;

    // scratch, one buffer per role so that calls can be chained without aliasing
    DevBuf rs_states, rs_scratch, rs_small, k6_scratch;  // the reference's random streams (dsac_refstream_init) and the scratch of a sampling window
    int rs_threads = 0, rs_mode = DSAC_RS_DEFAULT_MODE;
    int k6_walk_exact = 0;  // "k6_walk_exact": 1 = the split walk decides every cell by the fp64 residual (no fp32 filter; A/B)
    int k6_waves = 0;  // "k6_waves": waves per refinement problem of K6's walk (0 = by the problem count)
    DevBuf staged, staged_lo, staged_split, soft_part, bwd_staged, dRdH, grad_part, g12_part, g6;
    int g6_n = 0;  // hypotheses held by g6 (dsac_last_pose_gradients)
    // staging for host-pointer arguments: slots are bump-allocated per call.  A deque: next_slot() hands out references that
    // must stay valid while further slots are appended within the same call
    std::deque<DevBuf> slots;
    size_t slot_next = 0;
    struct Pending { void* host; const void* dev; size_t bytes; };
    std::vector<Pending> pending;
    dk::K2Opts k2;  // launch knobs, read once in dsac_create (DSAC_K2_*) or set with dsac_set_option; no process-wide state
    dk::K1Opts k1;
    int k1_cus = 0;       // > 0: the auxiliary stream of the pipelined pair (K1 of the next step) is confined to that many CUs (DSAC_K1_CUS / "k1_cus")
    bool device_args = false;  // "device_args": every pointer argument is a device pointer (the caller's promise; skips hipPointerGetAttributes per argument)
    int seed_stride = 1;  // frame f of a batch samples from the stream of seed + f * seed_stride ("seed_stride": images sharded round-robin over ranks keep their own seeds)
    int k4_variant = -1;  // K4 main-pass form (dk::backward_plan), DSAC_K4_VARIANT / dsac_set_option("k4_variant")
    hipEvent_t k2_wait = nullptr, k2_record = nullptr;  // optional gate around the bandwidth-bound kernel (dsac_set_k2_events)

    // two-slot software pipeline (dsac_sample_ahead / dsac_score_sampled): K1 of frame i+1 on `aux` under K2/K3 of frame i
    hipStream_t aux = nullptr, aux2 = nullptr;  // aux: K1 of the next frame; aux2: K3 tail of the previous frame
    DevBuf slot_staged[2], slot_soft_part[2];
    hipEvent_t slot_ready[2] = {nullptr, nullptr}, slot_free[2] = {nullptr, nullptr}, slot_reduced[2] = {nullptr, nullptr},
               slot_done[2] = {nullptr, nullptr};  // ready: K1 done; free: K2 done; reduced: partial sums consumed; done: K3 tail done
    bool slot_free_recorded[2] = {false, false}, slot_reduced_recorded[2] = {false, false}, slot_done_recorded[2] = {false, false};
    bool slot_pending[2] = {false, false};  // sampled, not yet scored
    int slot_N[2] = {0, 0};
    dk::FrameDev slot_F[2]{};               // the frame a slot was sampled from (its score call reprojects against the same one)
    hipEvent_t frame_ready = nullptr;       // recorded on `stream` after a frame copy; the auxiliary stream waits for it before K1

    // deferred refinement tail of dsac_process_images ("pi_defer_tail"): K6 / K7 of batch i on `tail` under K1 / K2 of batch i + 1
    int pi_defer_tail = 0;
    // two tail streams: mode 1 uses the first; mode 2 alternates (call i on stream i & 1) -- consecutive calls write different arrays there, so their
    // tails are independent, and in a loop of SINGLE images (where the one-wave refinement chain is longer than K1 + K2) two tails run side by side
    hipStream_t tail[2] = {nullptr, nullptr};
    hipEvent_t tail_go = nullptr, tail_done[2] = {nullptr, nullptr};  // go: K3 of the batch done (main stream); done: the tail's last launch done
    bool tail_pending[2] = {false, false};              // a tail is in flight on that stream that the main stream has not been ordered behind yet
    // mode 2 (score tail deferred as well: reduction + K3 of batch i also run on `tail`): the partial sums alternate between two buffers, and call
    // i + 2 starts only after K3 of call i (pi_scored: recorded on `tail`), which read that call's poses and partial sums
    DevBuf pi_soft[2], pi_scores[2];
    hipEvent_t pi_k2done = nullptr, pi_scored[2] = {nullptr, nullptr};
    bool pi_scored_rec[2] = {false, false};
    unsigned pi_calls = 0;
    int pi_tail_of[2] = {-1, -1};  // the tail stream the previous call of each parity used (mode 2)
    // dsac_process_images_begin ... dsac_process_images_finish: the pair shares one call parity (which half of the alternating arrays) and one size
    bool pi_open = false;
    int pi_open_b = 0, pi_open_N = 0, pi_open_frames = 0;
    int tail_prio = 0;             // DSAC_TAIL_PRIO / "tail_prio": create the tail streams with the highest stream priority (before the first deferred call)
    hipEvent_t xs_event = nullptr;                      // dsac_tail_wait: the context's stream as seen by another stream
    // dsac_backward_path1: K5 (dPNP needs nothing but the minimal sets) on a side stream beside the finite-difference replica chain of K6
    hipStream_t bwd_side = nullptr;
    hipEvent_t bwd_fork = nullptr, bwd_join = nullptr;

    // measurement hooks: event pairs around the dominant kernels
    bool profiling = false;
    int prof_stride = 1, prof_count[2] = {0, 0};  // record every prof_stride-th launch
    struct EvPair { hipEvent_t a, b; };
    std::vector<EvPair> ev[2];
    std::vector<EvPair> ev_free;
};

namespace {

int fail(dsac_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                                           \
    do {                                                                                                           \
        hipError_t e__ = (call);                                                                                   \
        if (e__ != hipSuccess) return fail((c), DSAC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__));      \
    } while (0)

// c: with dsac_set_option("device_args", 1) the caller has promised that every pointer argument is a device pointer: no runtime query
bool is_device_ptr(const void* p, const dsac_ctx* c = nullptr) {
    if (!p) return false;
    if (c && c->device_args) return true;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
}

// Order the main stream behind a deferred refinement tail that is still in flight (no-op otherwise).
void join_tail(dsac_ctx* c) {
    for (int k = 0; k < 2; k++) {
        if (!c->tail_pending[k]) continue;
        c->tail_pending[k] = false;
        if (hipStreamWaitEvent(c->stream, c->tail_done[k], 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->tail[k]); }
    }
}

// Begin a call: reset the staging slots.  Every call that enqueues work sees the results of a deferred tail in stream order; dsac_process_images
// itself keeps the tail in flight (keep_tail) and joins it right before the stage that would overwrite what the tail reads.
void begin_call(dsac_ctx* c, bool keep_tail = false) {
    c->slot_next = 0;
    c->pending.clear();
    if (!keep_tail) join_tail(c);
}

// The score tail of a scoring call on stream `st`: per-tile sums -> scores -> softmax / entropy / soft-argmax pose (two launches; a one-launch form with
// a ticket per frame was measured 4-5 us SLOWER per call -- its release / acquire fences write back and invalidate an L2 full of error images --)
hipError_t score_tail(hipStream_t st, int Nf, int frames, int tiles, const float* part, double* scores, double scale, double* w, double* ent,
                      const double* poses, double* avg) {
    hipError_t e = dk::reduce_soft(st, Nf * frames, tiles, part, scores);
    if (e != hipSuccess) return e;
    return dk::softmax(st, Nf, scores, scale, w, ent, poses, avg, frames);
}

DevBuf& next_slot(dsac_ctx* c) {
    if (c->slot_next >= c->slots.size()) c->slots.emplace_back();
    return c->slots[c->slot_next++];
}

// Input argument: returns a device pointer holding `bytes` of *p (copying if p is a host pointer).
template <typename T>
int in_arg(dsac_ctx* c, const T* p, size_t count, const T** out) {
    if (!p || count == 0) { *out = nullptr; return DSAC_OK; }
    if (is_device_ptr(p, c)) { *out = p; return DSAC_OK; }
    DevBuf& s = next_slot(c);
    HIP_TRY(c, s.reserve(count * sizeof(T)));
    HIP_TRY(c, hipMemcpyAsync(s.p, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *out = s.as<T>();
    return DSAC_OK;
}

// Output argument: device pointer to write into; host destinations are copied back by end_call.
// `preload` copies the current host contents up first (for accumulated outputs).
template <typename T>
int out_arg(dsac_ctx* c, T* p, size_t count, T** out, bool preload = false) {
    if (!p || count == 0) { *out = nullptr; return DSAC_OK; }
    if (is_device_ptr(p, c)) { *out = p; return DSAC_OK; }
    DevBuf& s = next_slot(c);
    HIP_TRY(c, s.reserve(count * sizeof(T)));
    if (preload) HIP_TRY(c, hipMemcpyAsync(s.p, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    c->pending.push_back({p, s.p, count * sizeof(T)});
    *out = s.as<T>();
    return DSAC_OK;
}

// End a call: copy pending host outputs back (synchronous only when there are any).
int end_call(dsac_ctx* c) {
    if (c->pending.empty()) return DSAC_OK;
    for (auto& pd : c->pending) HIP_TRY(c, hipMemcpyAsync(pd.host, pd.dev, pd.bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pending.clear();
    return DSAC_OK;
}

// RAII-ish helper: records an event pair around a launch when profiling is on.
// attached = true: the pair is handed to the launch (hipExtLaunchKernelGGL takes the kernel's own start / stop timestamps), nothing is recorded
// on the stream here.
struct ProfScope {
    dsac_ctx* c;
    int which;
    dsac_ctx::EvPair p{};
    bool on = false, attached = false, launched = false;  // launched: set by commit() once the kernel the pair is attached to really went out
    ProfScope(dsac_ctx* c_, int which_, bool attached_ = false) : c(c_), which(which_), attached(attached_) {
        if (!c->profiling) return;
        if ((c->prof_count[which]++ % c->prof_stride) != 0) return;
        if (!c->ev_free.empty()) { p = c->ev_free.back(); c->ev_free.pop_back(); }
        else if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        on = attached ? true : hipEventRecord(p.a, c->stream) == hipSuccess;
    }
    // the K2 options of this call: the context's, plus the event pair when this launch is sampled
    dk::K2Opts k2(const double* poses64 = nullptr) const {
        dk::K2Opts o = c->k2;
        if (on && attached) { o.ev_start = p.a; o.ev_stop = p.b; }
        o.poses64 = poses64;  // the cv poses of this launch: what the precise form ("k2_flags" bit 25) projects with
        o.staged_lo = (poses64 && (o.flags & dk::K2_FLAG_RECLO)) ? c->staged_lo.as<float>() : nullptr;  // filled by k2_records_lo() before the launch
        o.split = (poses64 && dk::k2_wants_exact(o) && dk::pose_split_available(c->F)) ? c->staged_split.as<char>() : nullptr;  // likewise
        return o;
    }
    void commit() { launched = true; }
    ~ProfScope() {
        if (!on) return;
        // an attached pair whose launch never happened (rejected variant, failed launch) was never recorded: reading it would fail every later
        // dsac_profile_read -- it goes back to the free list instead
        if (attached) { if (launched) c->ev[which].push_back(p); else c->ev_free.push_back(p); return; }
        if (hipEventRecord(p.b, c->stream) == hipSuccess) c->ev[which].push_back(p);
    }
};

// "k2_flags" bit 27: the low parts of the N staged records, derived from the cv poses on `st` right in front of the K2 launch that reads them
static hipError_t k2_records_lo(dsac_ctx* c, hipStream_t st, int N, const double* d_poses) {
    if (dk::k2_wants_exact(c->k2) && d_poses && N > 0 && dk::pose_split_available(c->F)) {  // the split records of the exact-transform form (the default; "k2_flags" bit 28)
        hipError_t e = c->staged_split.reserve(dk::pose_split_bytes(N));
        if (e != hipSuccess) return e;
        e = dk::pose_prep_split(st, N, d_poses, c->F, c->staged_split.as<char>());
        if (e != hipSuccess) return e;
    }
    if (!(c->k2.flags & dk::K2_FLAG_RECLO) || !d_poses || N <= 0) return hipSuccess;
    hipError_t e = c->staged_lo.reserve((size_t)N * dk::POSE_STRIDE * sizeof(float));
    if (e != hipSuccess) return e;
    return dk::pose_prep_lo(st, N, d_poses, c->F, c->staged_lo.as<float>());
}

#define ARG_TRY(expr)                \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != DSAC_OK) return rc__; \
    } while (0)

// dsac_gather_rows: row i of dst = row idx.r[i] of src, rows of `words` 32-bit words; blockIdx.y = destination row.  The indices travel as a kernel
// argument (no device buffer, no upload): up to 256 rows per launch.
struct GatherIdx { int32_t r[256]; };
template <typename V>
__global__ __launch_bounds__(256) void k_gather_rows(V* __restrict__ dst, const V* __restrict__ src, size_t vecs, GatherIdx idx) {
    const V* s = src + (size_t)idx.r[blockIdx.y] * vecs;
    V* d = dst + (size_t)blockIdx.y * vecs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ void k_quantise_int16(float* xyz, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // cv::saturate_cast<short>(float): round to nearest (even), then saturate
    float v = rintf(xyz[i]);
    v = fminf(fmaxf(v, -32768.0f), 32767.0f);
    xyz[i] = v;
}

}  // namespace

extern "C" {

const char* dsac_version(void) { return "dsac_hip 0.1 (gfx950)"; }

const char* dsac_last_error(dsac_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int dsac_create(dsac_ctx** out, int device) {
    if (!out) return fail(nullptr, DSAC_ERR_INVALID, "dsac_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(nullptr, DSAC_ERR_NO_DEVICE, "dsac_create: no HIP device (%s); this engine has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    }
    if (device < 0 || device >= count) return fail(nullptr, DSAC_ERR_INVALID, "dsac_create: device %d out of range [0,%d)", device, count);
    HIP_TRY(nullptr, hipSetDevice(device));
    dsac_ctx* c = new (std::nothrow) dsac_ctx();
    if (!c) return fail(nullptr, DSAC_ERR_ALLOC, "dsac_create: out of host memory");
    c->device = device;
    e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) { delete c; return fail(nullptr, DSAC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string n = c->prop.gcnArchName;
        delete c;
        return fail(nullptr, DSAC_ERR_NO_DEVICE, "dsac_create: device %d is %s; this library carries gfx950 code only", device, n.c_str());
    }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(nullptr, DSAC_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->own_stream = true;
    // experiment knobs, per context (dsac_set_option changes them later)
    if (const char* v = getenv("DSAC_K2_VARIANT")) c->k2.variant = atoi(v);
    if (const char* v = getenv("DSAC_TAIL_PRIO")) c->tail_prio = atoi(v) != 0;
    if (const char* v = getenv("DSAC_K2_ORDER")) c->k2.pixel_minor = atoi(v) != 0;
    if (const char* v = getenv("DSAC_K2_FLAGS")) c->k2.flags = atoi(v);
    if (const char* v = getenv("DSAC_K2_EXACT_AUTO")) c->k2.exact_auto = atoi(v) != 0;
    if (const char* v = getenv("DSAC_K1_WPB")) c->k1.wpb = atoi(v);
    if (const char* v = getenv("DSAC_K1_PRIO")) c->k1.prio = atoi(v);
    if (const char* v = getenv("DSAC_K1_HPW")) c->k1.hpw = atoi(v);
    if (const char* v = getenv("DSAC_K1_HORN")) c->k1.horn = atoi(v) != 0;
    if (const char* v = getenv("DSAC_K1_MINW")) c->k1.minw = atoi(v);
    if (const char* v = getenv("DSAC_K1_RL")) c->k1.rl = atoi(v) == 1 ? 1 : 4;
    if (const char* v = getenv("DSAC_K1_WIDE")) c->k1.wide = atoi(v);
    if (const char* v = getenv("DSAC_K1_SHARE")) { const int sv = atoi(v); c->k1.share = sv < 0 ? -sv : sv; c->k1.share_always = sv < 0; }
    if (const char* v = getenv("DSAC_K4_VARIANT")) c->k4_variant = atoi(v);
    if (const char* v = getenv("DSAC_K1_CUS")) c->k1_cus = atoi(v);
    *out = c;
    return DSAC_OK;
}

void dsac_destroy(dsac_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // drain every stream that may still read the context's buffers (a deferred tail runs K6 / K7 against the frame) before anything is freed
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < 2; k++)
        if (c->tail[k]) (void)hipStreamSynchronize(c->tail[k]);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->aux2) (void)hipStreamSynchronize(c->aux2);
    c->frame_xyz.release(); c->frame_uv.release();
    c->staged.release(); c->staged_lo.release(); c->staged_split.release(); c->rs_states.release(); c->rs_scratch.release(); c->rs_small.release(); c->k6_scratch.release(); c->soft_part.release(); c->bwd_staged.release(); c->dRdH.release();
    c->grad_part.release(); c->g12_part.release(); c->g6.release();
    for (auto& s : c->slots) s.release();
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->aux2) { (void)hipStreamSynchronize(c->aux2); (void)hipStreamDestroy(c->aux2); }
    for (int k = 0; k < 2; k++) {
        c->slot_staged[k].release();
        c->slot_soft_part[k].release();
        c->pi_soft[k].release();
        c->pi_scores[k].release();
        if (c->slot_reduced[k]) (void)hipEventDestroy(c->slot_reduced[k]);
        if (c->slot_ready[k]) (void)hipEventDestroy(c->slot_ready[k]);
        if (c->slot_free[k]) (void)hipEventDestroy(c->slot_free[k]);
        if (c->slot_done[k]) (void)hipEventDestroy(c->slot_done[k]);
    }
    if (c->frame_ready) (void)hipEventDestroy(c->frame_ready);
    for (int k = 0; k < 2; k++) {
        if (c->tail[k]) { (void)hipStreamSynchronize(c->tail[k]); (void)hipStreamDestroy(c->tail[k]); }
        if (c->tail_done[k]) (void)hipEventDestroy(c->tail_done[k]);
    }
    if (c->tail_go) (void)hipEventDestroy(c->tail_go);
    if (c->pi_k2done) (void)hipEventDestroy(c->pi_k2done);
    for (int k = 0; k < 2; k++)
        if (c->pi_scored[k]) (void)hipEventDestroy(c->pi_scored[k]);
    if (c->xs_event) (void)hipEventDestroy(c->xs_event);
    if (c->bwd_side) { (void)hipStreamSynchronize(c->bwd_side); (void)hipStreamDestroy(c->bwd_side); }
    if (c->bwd_fork) (void)hipEventDestroy(c->bwd_fork);
    if (c->bwd_join) (void)hipEventDestroy(c->bwd_join);
    for (int k = 0; k < 2; k++) for (auto& p : c->ev[k]) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : c->ev_free) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dsac_set_stream(dsac_ctx* c, void* hip_stream) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_set_stream: ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; k++)
        if (c->tail[k]) { HIP_TRY(c, hipStreamSynchronize(c->tail[k])); c->tail_pending[k] = false; }
    if (c->own_stream && c->stream) HIP_TRY(c, hipStreamDestroy(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return DSAC_OK;
}

void* dsac_get_stream(dsac_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }

int dsac_synchronize(dsac_ctx* c) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_synchronize: ctx is NULL");
    if (c->aux) HIP_TRY(c, hipStreamSynchronize(c->aux));
    if (c->aux2) HIP_TRY(c, hipStreamSynchronize(c->aux2));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; k++)
        if (c->tail[k]) { HIP_TRY(c, hipStreamSynchronize(c->tail[k])); c->tail_pending[k] = false; }
    return DSAC_OK;
}

int dsac_device_info(dsac_ctx* c, int* cus, int* clock_khz, uint64_t* mem_bytes, char* name64) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_device_info: ctx is NULL");
    if (cus) *cus = c->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = c->prop.clockRate;
    if (mem_bytes) *mem_bytes = (uint64_t)c->prop.totalGlobalMem;
    if (name64) { strncpy(name64, c->prop.gcnArchName, 63); name64[63] = 0; }
    return DSAC_OK;
}

static int set_frames_common(dsac_ctx* c, int frames, const float* xyz, const float* uv, bool uv_per_frame, int H, int W, float fx, float fy, float cx,
                             float cy, unsigned flags) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_set_frame: ctx is NULL");
    if (!xyz || H <= 0 || W <= 0 || frames <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_set_frame: need xyz, frames > 0 and H, W > 0 (got H=%d W=%d)", H, W);
    if ((long long)H * W * frames > (1ll << 30)) return fail(c, DSAC_ERR_INVALID, "dsac_set_frame: frames*H*W too large");
    if (!(fx != 0.f) || !(fy != 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_set_frame: zero focal length");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t P1 = (size_t)H * W;
    const size_t P = P1 * (size_t)frames;              // cells to copy for xyz
    const size_t Puv = uv_per_frame ? P : P1;
    const bool borrow = (flags & DSAC_FRAME_BORROW) != 0;
    if (!borrow && (c->slot_pending[0] || c->slot_pending[1]))
        return fail(c, DSAC_ERR_INVALID, "dsac_set_frame: a pipelined slot is sampled but not yet scored; the library's own frame copy cannot be replaced "
                                         "underneath it (use DSAC_FRAME_BORROW frames with the pipelined calls)");
    if (borrow) {
        if (!is_device_ptr(xyz, c) || (uv && !is_device_ptr(uv, c))) return fail(c, DSAC_ERR_INVALID, "dsac_set_frame: DSAC_FRAME_BORROW needs device pointers");
        if (flags & DSAC_FRAME_QUANTISE_INT16) {  // the caller's own device buffer, rounded to the int16 grid in place (core/cnn_softam.h:265)
            const size_t n = P * 3;
            hipLaunchKernelGGL(k_quantise_int16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, const_cast<float*>(xyz), n);
            HIP_TRY(c, hipGetLastError());
        }
        c->F.xyz = xyz;
        c->F.uv = uv;
    } else {
        join_tail(c);  // ... and so may a deferred refinement tail
        // K1 of an earlier pipelined slot (auxiliary stream) may still be reading the library's copy: order the overwrite behind it
        for (int k = 0; k < 2; k++)
            if (c->aux && c->slot_N[k] > 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->slot_ready[k], 0));
        HIP_TRY(c, c->frame_xyz.reserve(P * 3 * sizeof(float)));
        HIP_TRY(c, hipMemcpyAsync(c->frame_xyz.p, xyz, P * 3 * sizeof(float), hipMemcpyDefault, c->stream));
        c->F.xyz = c->frame_xyz.as<float>();
        if (uv) {
            HIP_TRY(c, c->frame_uv.reserve(Puv * 2 * sizeof(float)));
            HIP_TRY(c, hipMemcpyAsync(c->frame_uv.p, uv, Puv * 2 * sizeof(float), hipMemcpyDefault, c->stream));
            c->F.uv = c->frame_uv.as<float>();
        } else {
            c->F.uv = nullptr;
        }
        if (flags & DSAC_FRAME_QUANTISE_INT16) {
            const size_t n = P * 3;
            hipLaunchKernelGGL(k_quantise_int16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->frame_xyz.as<float>(), n);
            HIP_TRY(c, hipGetLastError());
        }
        if (!is_device_ptr(xyz) || (uv && !is_device_ptr(uv))) HIP_TRY(c, hipStreamSynchronize(c->stream));  // host source may be freed after return
        if (c->frame_ready) HIP_TRY(c, hipEventRecord(c->frame_ready, c->stream));  // the auxiliary stream's K1 waits for the copy
    }
    c->F.H = H; c->F.W = W; c->F.P = (int)P1;
    c->F.fx = fx; c->F.fy = fy; c->F.cx = cx; c->F.cy = cy;
    c->F.frames = frames;
    c->F.seed_stride = c->seed_stride;
    c->F.xyz_stride = (long long)P1 * 3;
    c->F.uv_stride = (uv && uv_per_frame) ? (long long)P1 * 2 : 0;
    c->have_frame = true;
    return DSAC_OK;
}

int dsac_device_alloc(dsac_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return fail(c, DSAC_ERR_INVALID, "dsac_device_alloc: NULL argument");
    *out = nullptr;
    if (bytes == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (hipMalloc(out, bytes) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(c, DSAC_ERR_ALLOC, "dsac_device_alloc: %zu bytes of device memory", bytes); }
    return DSAC_OK;
}

int dsac_device_free(dsac_ctx* c, void* p) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_device_free: ctx is NULL");
    if (!p) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipFree(p));  // waits for the work that still uses the buffer
    return DSAC_OK;
}

int dsac_host_alloc(dsac_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return fail(c, DSAC_ERR_INVALID, "dsac_host_alloc: NULL argument");
    *out = nullptr;
    if (bytes == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(c, DSAC_ERR_ALLOC, "dsac_host_alloc: %zu bytes of page-locked memory", bytes); }
    return DSAC_OK;
}

int dsac_host_free(dsac_ctx* c, void* p) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_host_free: ctx is NULL");
    if (!p) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipHostFree(p));
    return DSAC_OK;
}

int dsac_copy_async(dsac_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(c, DSAC_ERR_INVALID, "dsac_copy_async: NULL argument");
    if (bytes == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // While a deferred tail is in flight the copy is ordered behind it, whichever way it goes: a device SOURCE may be something the tail is still writing
    // (ref6 / out4 / steps_done / inlier maps), a device DESTINATION something it still reads (the frame's xyz / uv, gt, perm: an upload of the next
    // frame into the buffer the tail refines against would be a write-after-read race).  join_tail is a no-op without a pending tail.
    join_tail(c);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
    return DSAC_OK;
}

int dsac_gather_rows(dsac_ctx* c, void* dst, const void* src, size_t row_bytes, int n_rows, const int32_t* rows) {
    if (!c || (n_rows > 0 && (!dst || !src || !rows))) return fail(c, DSAC_ERR_INVALID, "dsac_gather_rows: NULL argument");
    if (n_rows <= 0 || row_bytes == 0) return DSAC_OK;
    if (row_bytes % 4 != 0) return fail(c, DSAC_ERR_INVALID, "dsac_gather_rows: row_bytes must be a multiple of 4");
    if (!is_device_ptr(dst, c) || !is_device_ptr(src, c)) return fail(c, DSAC_ERR_INVALID, "dsac_gather_rows: dst and src are device pointers");
    HIP_TRY(c, hipSetDevice(c->device));
    join_tail(c);  // a deferred refinement tail may still read what dst holds (the previous step's frames) or write what src holds
    const bool v16 = row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0;
    for (int r0 = 0; r0 < n_rows; r0 += 256) {
        const int n = n_rows - r0 < 256 ? n_rows - r0 : 256;
        GatherIdx idx;
        for (int i = 0; i < n; i++) {
            if (rows[r0 + i] < 0) return fail(c, DSAC_ERR_INVALID, "dsac_gather_rows: negative row index");
            idx.r[i] = rows[r0 + i];
        }
        char* d = static_cast<char*>(dst) + (size_t)r0 * row_bytes;
        if (v16) {
            const size_t vecs = row_bytes / 16;
            const unsigned gx = (unsigned)std::min<size_t>((vecs + 255) / 256, (size_t)std::max(1, 2048 / n));
            hipLaunchKernelGGL((k_gather_rows<uint4>), dim3(gx, (unsigned)n), dim3(256), 0, c->stream, reinterpret_cast<uint4*>(d), static_cast<const uint4*>(src), vecs, idx);
        } else {
            const size_t vecs = row_bytes / 4;
            const unsigned gx = (unsigned)std::min<size_t>((vecs + 255) / 256, (size_t)std::max(1, 2048 / n));
            hipLaunchKernelGGL((k_gather_rows<uint32_t>), dim3(gx, (unsigned)n), dim3(256), 0, c->stream, reinterpret_cast<uint32_t*>(d), static_cast<const uint32_t*>(src), vecs,
                               idx);
        }
        HIP_TRY(c, hipGetLastError());
    }
    return DSAC_OK;
}

int dsac_fill_zero_async(dsac_ctx* c, void* dst, size_t bytes) {
    if (!c || (bytes && !dst)) return fail(c, DSAC_ERR_INVALID, "dsac_fill_zero_async: NULL argument");
    if (bytes == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    join_tail(c);  // the buffer may be one a deferred tail still reads or writes (see dsac_copy_async)
    HIP_TRY(c, hipMemsetAsync(dst, 0, bytes, c->stream));
    return DSAC_OK;
}

int dsac_set_frame(dsac_ctx* c, const float* xyz, const float* uv, int H, int W, float fx, float fy, float cx, float cy, unsigned flags) {
    return set_frames_common(c, 1, xyz, uv, false, H, W, fx, fy, cx, cy, flags);
}

int dsac_set_frames(dsac_ctx* c, int frames, const float* xyz, const float* uv_or_null, int uv_per_frame, int H, int W, float fx, float fy, float cx,
                    float cy, unsigned flags) {
    return set_frames_common(c, frames, xyz, uv_or_null, uv_per_frame != 0, H, W, fx, fy, cx, cy, flags);
}

int dsac_sample(dsac_ctx* c, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, double* poses, int32_t* sets_out,
                uint8_t* ok) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_sample: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_sample: no frame set");
    // frame batch: N = frames x hypotheses per frame, hypothesis h samples frame h / (N / frames) from the stream of seed + frame * seed_stride
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (frames > 1 && (sets_or_null || N % frames != 0))
        return fail(c, DSAC_ERR_INVALID, "dsac_sample: with a frame batch N must be frames x (hypotheses per frame) and the sets are drawn here (given sets are "
                                         "evaluated frame by frame: dsac_set_frame + dsac_sample)");
    if (N < 0 || !poses || !sets_out || !ok) return fail(c, DSAC_ERR_INVALID, "dsac_sample: N >= 0 and poses/sets_out/ok must be non-NULL");
    if (N == 0) return DSAC_OK;
    if (!sets_or_null && max_tries <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_sample: max_tries must be > 0");
    if (!sets_or_null && c->F.P < 4) return fail(c, DSAC_ERR_INVALID, "dsac_sample: frame has fewer than 4 cells");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const int32_t* d_sets_in;
    double* d_poses;
    int32_t* d_sets_out;
    uint8_t* d_ok;
    ARG_TRY(in_arg(c, sets_or_null, (size_t)N * 4, &d_sets_in));
    ARG_TRY(out_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, sets_out, (size_t)N * 4, &d_sets_out));
    ARG_TRY(out_arg(c, ok, (size_t)N, &d_ok));
    HIP_TRY(c, dk::sample(c->stream, N, seed, d_sets_in, c->F, (int)thr, max_tries, d_poses, d_sets_out, d_ok, nullptr, frames > 1 ? N / frames : 0, c->k1));
    return end_call(c);
}

// ---- K1 in the reference's own random stream (core/thread_rand.cpp:40-69) -----------------------------------------------------------------
int dsac_refstream_init(dsac_ctx* c, unsigned seed, int threads) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refstream_init: ctx is NULL");
    if (threads < 1 || threads > 1024) return fail(c, DSAC_ERR_INVALID, "dsac_refstream_init: threads must be 1..1024");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    HIP_TRY(c, c->rs_states.reserve((size_t)threads * sizeof(dk::RefStreamState)));
    HIP_TRY(c, dk::refstream_init(c->stream, c->rs_states.as<dk::RefStreamState>(), seed, threads));
    c->rs_threads = threads;
    return DSAC_OK;
}

int dsac_refstream_discard(dsac_ctx* c, int thread, unsigned long long n32) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refstream_discard: ctx is NULL");
    if (c->rs_threads <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_refstream_discard: call dsac_refstream_init first");
    if (thread < 0 || thread >= c->rs_threads) return fail(c, DSAC_ERR_INVALID, "dsac_refstream_discard: thread %d of %d", thread, c->rs_threads);
    if (n32 > (1ull << 36)) return fail(c, DSAC_ERR_INVALID, "dsac_refstream_discard: at most 2^36 outputs per call");
    if (n32 == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    HIP_TRY(c, dk::refstream_discard(c->stream, c->rs_states.as<dk::RefStreamState>(), thread, n32));
    return DSAC_OK;
}

int dsac_sample_refstream(dsac_ctx* c, int N, float thr, long long max_attempts, double* poses, int32_t* sets_out, uint8_t* ok, unsigned long long* consumed32_or_null,
                          long long* attempts_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_sample_refstream: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_sample_refstream: no frame set");
    if (c->rs_threads <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_sample_refstream: call dsac_refstream_init first");
    if (c->F.frames > 1) return fail(c, DSAC_ERR_INVALID, "dsac_sample_refstream: one frame at a time (the reference's streams run through its images in sequence)");
    if (N < 0 || !poses || !sets_out || !ok) return fail(c, DSAC_ERR_INVALID, "dsac_sample_refstream: N >= 0 and poses/sets_out/ok must be non-NULL");
    if (max_attempts <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_sample_refstream: max_attempts (per stream) must be > 0");
    if (c->F.P < 4) return fail(c, DSAC_ERR_INVALID, "dsac_sample_refstream: frame has fewer than 4 cells");
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const int T = c->rs_threads;
    double* d_poses;
    int32_t* d_sets_out;
    uint8_t* d_ok;
    ARG_TRY(out_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, sets_out, (size_t)N * 4, &d_sets_out));
    ARG_TRY(out_arg(c, ok, (size_t)N, &d_ok));
    // per-stream bookkeeping on the device: first | served | need | parsed (int32 each), consumed (uint64), attempts (int64)
    const size_t small_bytes = (size_t)T * (4 * 4 + 8 + 8);
    HIP_TRY(c, c->rs_small.reserve(small_bytes));
    std::vector<char> hs(small_bytes, 0);
    int32_t* h_first = reinterpret_cast<int32_t*>(hs.data());
    int32_t* h_need = h_first + 2 * T;
    int max_need = 0;
    for (int t = 0; t < T; t++) {
        int f, n;
        rs::static_chunk(N, T, t, f, n);  // #pragma omp parallel for, static schedule (core/cnn_softam.h:1010)
        h_first[t] = f;
        h_need[t] = n;
        max_need = std::max(max_need, n);
    }
    char* ds = c->rs_small.as<char>();
    int32_t *d_first = reinterpret_cast<int32_t*>(ds), *d_served = d_first + T, *d_need = d_first + 2 * T, *d_parsed = d_first + 3 * T;
    unsigned long long* d_consumed = reinterpret_cast<unsigned long long*>(ds + (size_t)T * 16);
    long long* d_attempts = reinterpret_cast<long long*>(ds + (size_t)T * 24);
    HIP_TRY(c, hipMemcpyAsync(ds, hs.data(), small_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // hs is a local
    // windows: 16 attempts per wanted hypothesis to begin with (a frame with 70 % inliers accepts one attempt in ~20), doubling while a stream is unserved
    long long spent = 0;
    int A = 256;
    while (A < 16 * max_need && A < 16384) A *= 2;
    std::vector<int32_t> need_now(T);
    for (int round = 0; round < 4096; round++) {
        if ((long long)A > max_attempts - spent) A = (int)std::max(1ll, max_attempts - spent);
        HIP_TRY(c, c->rs_scratch.reserve(dk::refstream_window_bytes(T, A)));
        HIP_TRY(c, dk::refstream_window(c->stream, c->rs_states.as<dk::RefStreamState>(), T, A, c->rs_mode, c->rs_scratch.p, c->F, (int)thr, d_first, d_served, d_need, d_parsed,
                                        d_consumed, d_attempts, d_poses, d_sets_out, d_ok, nullptr));
        spent += A;
        HIP_TRY(c, hipMemcpyAsync(need_now.data(), d_need, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        int open = 0;
        for (int t = 0; t < T; t++) open = std::max(open, need_now[t]);
        if (open == 0 || spent >= max_attempts) break;
        if (A < 16384) A *= 2;
    }
    HIP_TRY(c, dk::refstream_unserved(c->stream, T, d_first, d_served, d_need, c->F, d_poses, d_sets_out, d_ok, nullptr));
    if (consumed32_or_null) HIP_TRY(c, hipMemcpyAsync(consumed32_or_null, d_consumed, (size_t)T * 8, hipMemcpyDefault, c->stream));
    if (attempts_or_null) HIP_TRY(c, hipMemcpyAsync(attempts_or_null, d_attempts, (size_t)T * 8, hipMemcpyDefault, c->stream));
    if (consumed32_or_null || attempts_or_null) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return end_call(c);
}

int dsac_reproject(dsac_ctx* c, int N, const double* poses, float clampv, float* err_or_null, float tau, float beta, double* soft_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_reproject: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_reproject: no frame set");
    // frame batch: N = frames x hypotheses per frame (a multiple of 128: no hypothesis tile of K2 straddles two frames), hypothesis h scores frame h / Nf
    const int Nf = c->F.frames > 1 ? N / c->F.frames : 0;
    if (c->F.frames > 1 && (N % c->F.frames != 0 || Nf % dk::K2_NF_MULTIPLE != 0))
        return fail(c, DSAC_ERR_INVALID, "dsac_reproject: with a frame batch N must be frames x (a multiple of %d), got %d for %d frames", dk::K2_NF_MULTIPLE, N, c->F.frames);
    if (N < 0 || !poses) return fail(c, DSAC_ERR_INVALID, "dsac_reproject: N >= 0 and poses must be non-NULL");
    if (N == 0 || (!err_or_null && !soft_or_null)) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double* d_poses;
    float* d_err;
    double* d_soft;
    ARG_TRY(in_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, err_or_null, (size_t)N * P, &d_err));
    ARG_TRY(out_arg(c, soft_or_null, (size_t)N, &d_soft));
    HIP_TRY(c, c->staged.reserve((size_t)N * dk::POSE_STRIDE * sizeof(float)));
    HIP_TRY(c, dk::pose_prep(c->stream, N, d_poses, c->F, c->staged.as<float>()));
    float* d_part = nullptr;
    const int tiles = dk::reproject_num_pixel_tiles(c->F.P);
    // Error images only on a big launch: the streaming form WITH the sigmoid arithmetic is the faster kernel -- the arithmetic spaces a wave's stores
    // (N = 4096: 860 us against 890-900 for any form without it; idling the wave instead, k2_flags bits 16-23, does not reproduce the effect:
    // profiles/r04_k2_err_ab.txt).  So the auto policy runs that kernel and drops its partial sums (1200 x N floats of scratch, 0.4 % of the
    // stores; no reduction launch).  k2_flags bit 24 switches the policy off (A/B).
    const bool fused_for_err = !d_soft && d_err && c->k2.variant < 0 && !(c->k2.flags & (1 << 24)) && (double)N * (double)P * 4.0 > 1.0e9;
    if (d_soft || fused_for_err) {
        HIP_TRY(c, c->soft_part.reserve((size_t)tiles * N * sizeof(float)));
        d_part = c->soft_part.as<float>();
    }
    if (fused_for_err && !(beta > 0.f)) { tau = 10.f; beta = 0.5f; }  // any finite sigmoid; the sums are not used
    int used = 0;
    if (c->k2_wait) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->k2_wait, 0));
    {
        HIP_TRY(c, k2_records_lo(c, c->stream, N, d_poses));
        ProfScope ps(c, 0, true);
        HIP_TRY(c, dk::reproject(c->stream, N, c->staged.as<float>(), c->F, clampv, d_err, tau, beta, d_part, ps.k2(d_poses), &used, Nf));
        ps.commit();
    }
    if (c->k2_record) HIP_TRY(c, hipEventRecord(c->k2_record, c->stream));
    if (d_soft) HIP_TRY(c, dk::reduce_soft(c->stream, N, used, d_part, d_soft));
    return end_call(c);
}

static int softmax_common(dsac_ctx* c, const char* who, int frames, int N, const double* scores, double scale, double* w, double* entropy_or_null,
                          const double* poses_or_null, double* avg6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "%s: ctx is NULL", who);
    if (N <= 0 || frames <= 0 || !scores || !w) return fail(c, DSAC_ERR_INVALID, "%s: N > 0, frames > 0 and scores/w must be non-NULL", who);
    if ((long long)N * frames > (1ll << 24)) return fail(c, DSAC_ERR_INVALID, "%s: too many scores", who);
    if ((avg6_or_null != nullptr) != (poses_or_null != nullptr)) return fail(c, DSAC_ERR_INVALID, "%s: avg6 and poses go together", who);
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t NT = (size_t)N * frames;
    const double *d_scores, *d_poses;
    double *d_w, *d_ent, *d_avg;
    ARG_TRY(in_arg(c, scores, NT, &d_scores));
    ARG_TRY(in_arg(c, poses_or_null, NT * 6, &d_poses));
    ARG_TRY(out_arg(c, w, NT, &d_w));
    ARG_TRY(out_arg(c, entropy_or_null, (size_t)frames, &d_ent));
    ARG_TRY(out_arg(c, avg6_or_null, (size_t)frames * 6, &d_avg));
    HIP_TRY(c, dk::softmax(c->stream, N, d_scores, scale, d_w, d_ent, d_poses, d_avg, frames));
    return end_call(c);
}

int dsac_softmax(dsac_ctx* c, int N, const double* scores, double scale, double* w, double* entropy_or_null, const double* poses_or_null,
                 double* avg6_or_null) {
    return softmax_common(c, "dsac_softmax", 1, N, scores, scale, w, entropy_or_null, poses_or_null, avg6_or_null);
}

int dsac_softmax_frames(dsac_ctx* c, int frames, int hyps_per_frame, const double* scores, double scale, double* w, double* entropy_or_null,
                        const double* poses_or_null, double* avg6_or_null) {
    return softmax_common(c, "dsac_softmax_frames", frames, hyps_per_frame, scores, scale, w, entropy_or_null, poses_or_null, avg6_or_null);
}

static int pi_tail_index(int mode, int b, long long N, long long P);
static int pi_tail_setup(dsac_ctx* c, int mode, int tk);

static int score_hypotheses_common(dsac_ctx* c, int N, int Nf, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, float clampv, float tau,
                                   float beta, double scale, double* poses, int32_t* sets_out, uint8_t* ok, float* err_or_null, double* scores_or_null,
                                   double* w, double* entropy_or_null, double* avg6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_score_hypotheses: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_score_hypotheses: no frame set");
    if (N <= 0 || !poses || !sets_out || !ok || !w) return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses: N > 0 and poses/sets_out/ok/w must be non-NULL");
    const int frames = Nf > 0 ? N / Nf : 1;
    if (!sets_or_null && max_tries <= 0) return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses: max_tries must be > 0");
    if (!sets_or_null && c->F.P < 4) return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses: frame has fewer than 4 cells");
    HIP_TRY(c, hipSetDevice(c->device));
    // "pi_defer_tail" = 2 on a frame batch with device-resident arguments: the score tail (reduction of the per-tile sums + K3: two small launches that
    // leave the chip idle while they run in order) goes to the tail stream, K1 of the NEXT call follows K2 of this one without a gap -- the contract of
    // dsac_process_images' mode 2: scores / w / entropy / avg6 (and poses / sets / ok, which K3 reads) are complete after dsac_join_tail / another entry
    // point / dsac_synchronize, consecutive calls are given different arrays (the error images may be the same buffer: only K2 touches them)
    const bool want_defer = c->pi_defer_tail == 2 && !sets_or_null;  // frame batches and (since the single-frame loop of configs[1] pays the same tail) one frame alike
    begin_call(c, /*keep_tail=*/want_defer);
    const size_t P = (size_t)c->F.P;
    const int32_t* d_sets_in;
    double *d_poses, *d_scores, *d_w, *d_ent, *d_avg;
    int32_t* d_sets_out;
    uint8_t* d_ok;
    float* d_err;
    ARG_TRY(in_arg(c, sets_or_null, (size_t)N * 4, &d_sets_in));
    ARG_TRY(out_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, sets_out, (size_t)N * 4, &d_sets_out));
    ARG_TRY(out_arg(c, ok, (size_t)N, &d_ok));
    ARG_TRY(out_arg(c, err_or_null, (size_t)N * P, &d_err));
    ARG_TRY(out_arg(c, scores_or_null, (size_t)N, &d_scores));
    ARG_TRY(out_arg(c, w, (size_t)N, &d_w));
    ARG_TRY(out_arg(c, entropy_or_null, (size_t)frames, &d_ent));
    ARG_TRY(out_arg(c, avg6_or_null, (size_t)frames * 6, &d_avg));
    const bool defer = want_defer && c->pending.empty();  // every output in HBM: nothing to copy back at the end of this call
    if (want_defer && !defer) join_tail(c);
    const int b = defer ? (int)(c->pi_calls++ & 1u) : 0;
    if (defer) c->pi_open = false;
    if (!d_scores) {
        if (defer) {
            HIP_TRY(c, c->pi_scores[b].reserve((size_t)N * sizeof(double)));
            d_scores = c->pi_scores[b].as<double>();
        } else {
            DevBuf& s = next_slot(c);
            HIP_TRY(c, s.reserve((size_t)N * sizeof(double)));
            d_scores = s.as<double>();
        }
    }
    const int tiles = dk::reproject_num_pixel_tiles(c->F.P);
    HIP_TRY(c, c->staged.reserve((size_t)N * dk::POSE_STRIDE * sizeof(float)));
    if (defer) {
        DevBuf& part = c->pi_soft[b];
        HIP_TRY(c, part.reserve((size_t)tiles * N * sizeof(float)));
        const int tk = pi_tail_index(2, b, N, (long long)P);
        ARG_TRY(pi_tail_setup(c, 2, tk));
        if (c->pi_scored_rec[b]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pi_scored[b], 0));  // K3 of the call two back read this half's arrays
        HIP_TRY(c, dk::sample(c->stream, N, seed, nullptr, c->F, (int)thr, max_tries, d_poses, d_sets_out, d_ok, c->staged.as<float>(), Nf, c->k1));
        int used_d = 0;
        if (c->k2_wait) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->k2_wait, 0));
        hipEvent_t k2_done = nullptr;
        {
            HIP_TRY(c, k2_records_lo(c, c->stream, N, d_poses));
            ProfScope ps(c, 0, true);
            dk::K2Opts o = ps.k2(d_poses);
            if (!o.ev_stop) o.ev_stop = c->pi_k2done;  // the tail's start rides on K2's own dispatch packet: no record between K2 and the next K1
            k2_done = o.ev_stop;
            HIP_TRY(c, dk::reproject(c->stream, N, c->staged.as<float>(), c->F, clampv, d_err, tau, beta, part.as<float>(), o, &used_d, Nf));
            ps.commit();
        }
        if (c->k2_record) HIP_TRY(c, hipEventRecord(c->k2_record, c->stream));
        hipStream_t ts = c->tail[tk];
        if (c->pi_tail_of[b] >= 0 && c->pi_tail_of[b] != tk) HIP_TRY(c, hipStreamWaitEvent(ts, c->tail_done[c->pi_tail_of[b]], 0));
        c->pi_tail_of[b] = tk;
        HIP_TRY(c, hipStreamWaitEvent(ts, k2_done, 0));
        HIP_TRY(c, score_tail(ts, Nf > 0 ? Nf : N, frames, used_d, part.as<float>(), d_scores, scale, d_w, d_ent, avg6_or_null ? d_poses : nullptr, d_avg));
        HIP_TRY(c, hipEventRecord(c->pi_scored[b], ts));
        c->pi_scored_rec[b] = true;
        HIP_TRY(c, hipEventRecord(c->tail_done[tk], ts));
        c->tail_pending[tk] = true;
        return DSAC_OK;
    }
    HIP_TRY(c, c->soft_part.reserve((size_t)tiles * N * sizeof(float)));
    // K1 writes the poses AND their staged K2 records (no separate pose_prep launch)
    HIP_TRY(c, dk::sample(c->stream, N, seed, d_sets_in, c->F, (int)thr, max_tries, d_poses, d_sets_out, d_ok, c->staged.as<float>(), Nf, c->k1));
    int used = 0;
    if (c->k2_wait) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->k2_wait, 0));
    {
        HIP_TRY(c, k2_records_lo(c, c->stream, N, d_poses));
        ProfScope ps(c, 0, true);
        HIP_TRY(c, dk::reproject(c->stream, N, c->staged.as<float>(), c->F, clampv, d_err, tau, beta, c->soft_part.as<float>(), ps.k2(d_poses), &used, Nf));
        ps.commit();
    }
    if (c->k2_record) HIP_TRY(c, hipEventRecord(c->k2_record, c->stream));
    HIP_TRY(c, score_tail(c->stream, Nf > 0 ? Nf : N, frames, used, c->soft_part.as<float>(), d_scores, scale, d_w, d_ent, avg6_or_null ? d_poses : nullptr,
                          d_avg));
    return end_call(c);
}

int dsac_score_hypotheses(dsac_ctx* c, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, float clampv, float tau,
                          float beta, double scale, double* poses, int32_t* sets_out, uint8_t* ok, float* err_or_null, double* scores_or_null,
                          double* w, double* entropy_or_null, double* avg6_or_null) {
    if (c && c->have_frame && c->F.frames > 1) return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses: a frame batch is set (use dsac_score_hypotheses_frames)");
    return score_hypotheses_common(c, N, 0, seed, sets_or_null, thr, max_tries, clampv, tau, beta, scale, poses, sets_out, ok, err_or_null, scores_or_null, w,
                                   entropy_or_null, avg6_or_null);
}

int dsac_score_hypotheses_frames(dsac_ctx* c, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clampv, float tau, float beta, double scale,
                                 double* poses, int32_t* sets_out, uint8_t* ok, float* err_or_null, double* scores_or_null, double* w,
                                 double* entropy_or_null, double* avg6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_score_hypotheses_frames: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_score_hypotheses_frames: no frame set");
    if (hyps_per_frame <= 0 || hyps_per_frame % 128 != 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses_frames: hyps_per_frame must be a positive multiple of 128 (got %d)", hyps_per_frame);
    if ((long long)hyps_per_frame * c->F.frames > (1ll << 24)) return fail(c, DSAC_ERR_INVALID, "dsac_score_hypotheses_frames: too many hypotheses");
    return score_hypotheses_common(c, hyps_per_frame * c->F.frames, hyps_per_frame, seed, nullptr, thr, max_tries, clampv, tau, beta, scale, poses, sets_out, ok,
                                   err_or_null, scores_or_null, w, entropy_or_null, avg6_or_null);
}

static int pipeline_init(dsac_ctx* c) {
    if (c->aux) return DSAC_OK;
    if (c->k1_cus > 0 && c->k1_cus < c->prop.multiProcessorCount) {
        // Confine the latency-bound sampling kernel to a few CUs: 2048 one-wave solves then take about as long as the bandwidth-bound K2 of
        // the previous step, which keeps (almost) the whole chip -- instead of K1's 288-register waves displacing K2's on every SIMD.
        const int ncu = c->prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int k = 0; k < c->k1_cus; k++) {  // evenly spread over the CU index space (every XCD contributes)
            const int cu = (int)(((long long)k * ncu) / c->k1_cus);
            mask[cu >> 5] |= 1u << (cu & 31);
        }
        HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->aux, (uint32_t)words, mask.data()));
    } else
    HIP_TRY(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->slot_ready[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->slot_free[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->slot_reduced[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->slot_done[k], hipEventDisableTiming));
    }
    HIP_TRY(c, hipEventCreateWithFlags(&c->frame_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->frame_ready, c->stream));  // covers a frame copy enqueued before the pipeline existed
    return DSAC_OK;
}

int dsac_sample_ahead(dsac_ctx* c, int slot, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, double* poses,
                      int32_t* sets_out, uint8_t* ok) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_sample_ahead: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_sample_ahead: no frame set");
    if (slot < 0 || slot > 1 || N <= 0 || !poses || !sets_out || !ok) return fail(c, DSAC_ERR_INVALID, "dsac_sample_ahead: slot in {0,1}, N > 0, non-NULL outputs");
    int Nf = 0;  // frame batch: N = frames x hypotheses per frame, frame f draws from the stream of seed + f
    if (c->F.frames > 1) {
        if (sets_or_null || N % c->F.frames != 0 || (N / c->F.frames) % 128 != 0)
            return fail(c, DSAC_ERR_INVALID, "dsac_sample_ahead: with a frame batch N must be frames x (a multiple of 128) and the sets are drawn here");
        Nf = N / c->F.frames;
    }
    if (!is_device_ptr(poses) || !is_device_ptr(sets_out) || !is_device_ptr(ok) || (sets_or_null && !is_device_ptr(sets_or_null)))
        return fail(c, DSAC_ERR_INVALID, "dsac_sample_ahead: the pipelined calls need device pointers");
    if (!sets_or_null && (max_tries <= 0 || c->F.P < 4)) return fail(c, DSAC_ERR_INVALID, "dsac_sample_ahead: max_tries > 0 and >= 4 cells needed");
    // every rejection comes before anything is touched: a protocol-violating call must leave the slot (its staged poses, K1's output) as it was
    if (c->slot_pending[slot]) return fail(c, DSAC_ERR_INVALID, "dsac_sample_ahead: slot %d is sampled but not yet scored", slot);
    join_tail(c);
    HIP_TRY(c, hipSetDevice(c->device));
    ARG_TRY(pipeline_init(c));
    HIP_TRY(c, c->slot_staged[slot].reserve((size_t)N * dk::POSE_STRIDE * sizeof(float)));
    if (c->slot_free_recorded[slot]) HIP_TRY(c, hipStreamWaitEvent(c->aux, c->slot_free[slot], 0));  // K2 of the slot's previous use has read its staged poses
    // ... and the K3 tail of that use (second auxiliary stream) has read the caller's `poses` for the soft-argmax average
    if (c->slot_done_recorded[slot]) HIP_TRY(c, hipStreamWaitEvent(c->aux, c->slot_done[slot], 0));
    HIP_TRY(c, hipStreamWaitEvent(c->aux, c->frame_ready, 0));  // a frame copy enqueued on the main stream has landed
    c->slot_F[slot] = c->F;  // the slot scores against the frame it was sampled from, whatever is current at score time
    HIP_TRY(c, dk::sample(c->aux, N, seed, sets_or_null, c->F, (int)thr, max_tries, poses, sets_out, ok, c->slot_staged[slot].as<float>(), Nf, c->k1));
    HIP_TRY(c, hipEventRecord(c->slot_ready[slot], c->aux));
    c->slot_N[slot] = N;
    c->slot_pending[slot] = true;
    return DSAC_OK;
}

int dsac_score_sampled(dsac_ctx* c, int slot, float clampv, float tau, float beta, double scale, const double* poses, float* err_or_null,
                       double* scores, double* w, double* entropy_or_null, double* avg6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_score_sampled: ctx is NULL");
    if (slot < 0 || slot > 1 || !c->aux || c->slot_N[slot] <= 0 || !c->slot_pending[slot])
        return fail(c, DSAC_ERR_INVALID, "dsac_score_sampled: no dsac_sample_ahead pending on this slot");
    if (!scores || !w || (avg6_or_null && !poses)) return fail(c, DSAC_ERR_INVALID, "dsac_score_sampled: scores/w (and poses with avg6) must be non-NULL");
    if (!is_device_ptr(scores) || !is_device_ptr(w) || (err_or_null && !is_device_ptr(err_or_null)) || (poses && !is_device_ptr(poses)) ||
        (entropy_or_null && !is_device_ptr(entropy_or_null)) || (avg6_or_null && !is_device_ptr(avg6_or_null)))
        return fail(c, DSAC_ERR_INVALID, "dsac_score_sampled: the pipelined calls need device pointers");
    join_tail(c);
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->slot_N[slot];
    const dk::FrameDev& SF = c->slot_F[slot];
    const int frames = SF.frames > 1 ? SF.frames : 1;
    const int Nf = frames > 1 ? N / frames : 0;
    const int tiles = dk::reproject_num_pixel_tiles(SF.P);
    HIP_TRY(c, c->slot_soft_part[slot].reserve((size_t)tiles * N * sizeof(float)));
    float* part = c->slot_soft_part[slot].as<float>();
    // main stream: nothing but the bandwidth-bound kernel, back to back
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->slot_ready[slot], 0));  // normally long satisfied: K1 ran under the previous K2
    if (c->slot_reduced_recorded[slot]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->slot_reduced[slot], 0));  // partials of frame i-2 consumed
    int used = 0;
    {
        HIP_TRY(c, k2_records_lo(c, c->stream, N, poses));
        ProfScope ps(c, 0, true);
        HIP_TRY(c, dk::reproject(c->stream, N, c->slot_staged[slot].as<float>(), SF, clampv, err_or_null, tau, beta, part, ps.k2(poses), &used, Nf));
        ps.commit();
    }
    HIP_TRY(c, hipEventRecord(c->slot_free[slot], c->stream));
    c->slot_free_recorded[slot] = true;
    c->slot_pending[slot] = false;
    // second auxiliary stream: the small latency-bound tail (partial sums -> scores -> softmax) runs under the next K2
    HIP_TRY(c, hipStreamWaitEvent(c->aux2, c->slot_free[slot], 0));
    HIP_TRY(c, dk::reduce_soft(c->aux2, N, used, part, scores));
    HIP_TRY(c, hipEventRecord(c->slot_reduced[slot], c->aux2));
    c->slot_reduced_recorded[slot] = true;
    HIP_TRY(c, dk::softmax(c->aux2, frames > 1 ? Nf : N, scores, scale, w, entropy_or_null, avg6_or_null ? poses : nullptr, avg6_or_null, frames));
    HIP_TRY(c, hipEventRecord(c->slot_done[slot], c->aux2));  // `poses` of this slot may be overwritten by the next dsac_sample_ahead
    c->slot_done_recorded[slot] = true;
    return DSAC_OK;
}

int dsac_dpnp(dsac_ctx* c, int N, const int32_t* sets, float eps, double* J) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_dpnp: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_dpnp: no frame set");
    if (c->F.frames > 1 && N % c->F.frames != 0) return fail(c, DSAC_ERR_INVALID, "dsac_dpnp: with a frame batch N must be frames x (sets per frame), got %d for %d frames", N, c->F.frames);
    if (N < 0 || !sets || !J || !(eps > 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_dpnp: need sets, J and eps > 0");
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const int32_t* d_sets;
    double* d_J;
    ARG_TRY(in_arg(c, sets, (size_t)N * 4, &d_sets));
    ARG_TRY(out_arg(c, J, (size_t)N * 72, &d_J));
    HIP_TRY(c, dk::dpnp(c->stream, N, d_sets, c->F, eps, d_J, c->F.frames > 1 ? N / c->F.frames : 0));
    return end_call(c);
}

int dsac_set_option(dsac_ctx* c, const char* key, int value) {
    if (!c || !key) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: NULL argument");
    const std::string k = key;
    if (k == "k2_variant") {
        if (!dk::reproject_variant_known(value)) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: unknown K2 kernel form %d", value);
        c->k2.variant = value;
    }
    else if (k == "k2_order") c->k2.pixel_minor = value != 0;
    else if (k == "k2_flags") c->k2.flags = value;
    else if (k == "k2_diag") c->k2.diag = value;
    else if (k == "k2_exact_auto") c->k2.exact_auto = value != 0;
    else if (k == "k6_walk_exact") c->k6_walk_exact = value != 0;
    else if (k == "k6_scan_tune") {  // A/B of the scan (process-wide): problems per wave | chunk cells << 8 | no skip check << 24
        const int g = value & 255, chunk = (value >> 8) & 0xffff;
        if ((g != 0 && g != 1 && g != 2 && g != 4) || chunk % 256 != 0 || (value >> 25) != 0) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: k6_scan_tune is g | chunk << 8 | noskip << 24 with g in {0, 1, 2, 4} and chunk a multiple of 256");
        dk::refine_scan_tune(value);
    }
    else if (k == "k6_waves") {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: k6_waves is 0 (auto), 1, 2, 4 or 8");
        c->k6_waves = value;
    }
    else if (k == "refstream_mode") {
        if (value < -1 || value > 1) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: refstream_mode is 0 (libstdc++ >= 11), 1 (libstdc++ <= 10) or -1 (this build's)");
        c->rs_mode = value < 0 ? DSAC_RS_DEFAULT_MODE : value;
    }
    else if (k == "k1_wpb") c->k1.wpb = value;
    else if (k == "k1_prio") c->k1.prio = value;
    else if (k == "k1_hpw") c->k1.hpw = value;
    else if (k == "k1_horn") c->k1.horn = value != 0;
    else if (k == "k1_minw") c->k1.minw = value;
    else if (k == "k1_rl") c->k1.rl = value == 1 ? 1 : 4;
    else if (k == "k1_wide") c->k1.wide = value;
    else if (k == "k1_share") { c->k1.share = value < 0 ? -value : value; c->k1.share_always = value < 0; }
    else if (k == "k4_variant") {
        if (!dk::backward_variant_known(value)) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: unknown K4 kernel form %d", value);
        c->k4_variant = value;
    }
    else if (k == "device_args") c->device_args = value != 0;
    else if (k == "tail_prio") {
        if (c->tail[0] || c->tail[1]) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: tail_prio must be set before the first deferred dsac_process_images");
        c->tail_prio = value != 0;
    }
    else if (k == "seed_stride") {
        if (value < 1) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: seed_stride must be >= 1");
        c->seed_stride = value;
        c->F.seed_stride = value;
    }
    else if (k == "pi_defer_tail") {
        // a CHANGE of mode orders the stream behind a tail in flight; setting the mode it already has is a no-op (a host that sets it before every
        // batch must not serialise the batches: the C++ FrameBatch did, and its tails sat exposed -- 70 us per image instead of 65)
        if (value < 0 || value > 2) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: pi_defer_tail is 0 (in order), 1 (refinement tail deferred) or 2 (score tail too)");
        if (value != c->pi_defer_tail) { join_tail(c); c->pi_defer_tail = value; }
    }
    else if (k == "k1_cus") { if (c->aux) return fail(c, DSAC_ERR_INVALID, "dsac_set_option: k1_cus must be set before the first dsac_sample_ahead"); c->k1_cus = value; }
    else return fail(c, DSAC_ERR_INVALID, "dsac_set_option: unknown key '%s'", key);
    return DSAC_OK;
}

int dsac_set_k2_events(dsac_ctx* c, void* wait_before_or_null, void* record_after_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_set_k2_events: ctx is NULL");
    c->k2_wait = reinterpret_cast<hipEvent_t>(wait_before_or_null);
    c->k2_record = reinterpret_cast<hipEvent_t>(record_after_or_null);
    return DSAC_OK;
}

int dsac_profile_enable(dsac_ctx* c, int on) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_profile_enable: ctx is NULL");
    c->profiling = on != 0;
    c->prof_stride = on > 1 ? on : 1;  // on = n > 1: time every n-th launch only (event records are not free on the stream)
    c->prof_count[0] = c->prof_count[1] = 0;
    return DSAC_OK;
}

int dsac_profile_read(dsac_ctx* c, int which, double* ms_total, int* launches, int reset) {
    if (!c || which < 0 || which > 1) return fail(c, DSAC_ERR_INVALID, "dsac_profile_read: bad arguments");
    double total = 0;
    int n = 0;
    for (auto& p : c->ev[which]) {
        HIP_TRY(c, hipEventSynchronize(p.b));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
        total += ms;
        n++;
    }
    if (ms_total) *ms_total = total;
    if (launches) *launches = n;
    if (reset) {
        for (auto& p : c->ev[which]) c->ev_free.push_back(p);
        c->ev[which].clear();
    }
    return DSAC_OK;
}

static int score_backward_common(dsac_ctx* c, const char* who, int N, const double* poses, const int32_t* sets, const float* d_err,
                                 const double* g, float clampv, float tau, float beta, const double* dpnp_or_null, unsigned flags,
                                 double* grad_xyz) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "%s: ctx is NULL", who);
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "%s: no frame set", who);
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    const int Nf = frames > 1 ? N / frames : 0;
    // one launch for all frames needs 16 | hypotheses per frame <= 256; any other count, and the fp64 parity mode, run frame by frame below
    if (frames > 1 && (N % frames != 0 || Nf <= 0))
        return fail(c, DSAC_ERR_INVALID, "%s: with a frame batch N must be frames x (hypotheses per frame), got %d for %d frames", who, N, frames);
    // the reference's jp-convention Jacobians use one focal length, f = camMat(0,0), for both axes (core/cnn_softam.h:406,466);
    // a camera with fx != fy would make this backward inconsistent with the forward kernels, which honour both
    if (c->F.fx != c->F.fy) return fail(c, DSAC_ERR_INVALID, "%s: needs fx == fy (got %g, %g): dProjectdObj / dProjectdHyp use a single focal length", who,
                                        (double)c->F.fx, (double)c->F.fy);
    if (N < 0 || !poses || !sets || !grad_xyz || (!d_err && !g)) return fail(c, DSAC_ERR_INVALID, "%s: NULL argument", who);
    if ((flags & DSAC_BWD_QUIRK_TRANSPOSE) && c->F.H != c->F.W)
        return fail(c, DSAC_ERR_INVALID, "%s: DSAC_BWD_QUIRK_TRANSPOSE needs a square map (H=%d W=%d)", who, c->F.H, c->F.W);
    const bool parity = (flags & DSAC_BWD_PARITY_FP64) != 0;
    if ((flags & DSAC_BWD_QUIRK_ROT_WRITEBACK) && !parity)
        return fail(c, DSAC_ERR_INVALID, "%s: DSAC_BWD_QUIRK_ROT_WRITEBACK needs DSAC_BWD_PARITY_FP64 (the write-back is a sequential recurrence)", who);
    if (parity && (!d_err || (long long)(frames > 1 ? Nf : N) * c->F.P > (1ll << 26)))
        return fail(c, DSAC_ERR_INVALID, "%s: DSAC_BWD_PARITY_FP64 takes a d_err volume with N*H*W <= 2^26 (reference-sized maps)", who);
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double *d_poses, *d_dpnp, *d_g;
    const int32_t* d_sets;
    const float* d_derr;
    double* d_grad;
    ARG_TRY(in_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(in_arg(c, sets, (size_t)N * 4, &d_sets));
    ARG_TRY(in_arg(c, d_err, (size_t)N * P, &d_derr));
    ARG_TRY(in_arg(c, g, (size_t)N, &d_g));
    ARG_TRY(in_arg(c, dpnp_or_null, (size_t)N * 72, &d_dpnp));
    ARG_TRY(out_arg(c, grad_xyz, (size_t)frames * P * 3, &d_grad, /*preload=*/true));
    if (!d_dpnp) {  // dPNP with the reference's default eps (core/cnn_softam.h:104)
        DevBuf& s = next_slot(c);
        HIP_TRY(c, s.reserve((size_t)N * 72 * sizeof(double)));
        HIP_TRY(c, dk::dpnp(c->stream, N, d_sets, c->F, 0.1f, s.as<double>(), Nf));
        d_dpnp = s.as<double>();
    }
    auto frame_view = [&](int f) {  // frame f of a batch as a single frame
        dk::FrameDev Fd = c->F;
        Fd.frames = 1;
        Fd.xyz = c->F.xyz + (size_t)f * c->F.xyz_stride;
        if (Fd.uv) Fd.uv = c->F.uv + (size_t)f * c->F.uv_stride;
        Fd.xyz_stride = Fd.uv_stride = 0;
        return Fd;
    };
    if (parity) {
        // fp64 parity mode: the reference's evaluation order, optional rotation write-back (quirk 7); a batch frame by frame (the scratch reused in stream order)
        const int n1 = frames > 1 ? Nf : N;
        DevBuf& jac = next_slot(c);
        HIP_TRY(c, jac.reserve((size_t)n1 * P * 3 * sizeof(double)));
        HIP_TRY(c, c->g6.reserve((size_t)N * 6 * sizeof(double)));
        for (int f = 0; f < frames; f++) {
            const size_t h0 = (size_t)f * n1;
            HIP_TRY(c, dk::score_backward_parity(c->stream, n1, d_poses + h0 * 6, frames > 1 ? frame_view(f) : c->F, d_derr + h0 * P, d_dpnp + h0 * 72, d_sets + h0 * 4, flags,
                                                 jac.as<double>(), d_grad + (size_t)f * P * 3, c->g6.as<double>() + h0 * 6));
        }
        c->g6_n = N;
        return end_call(c);
    }
    dk::K4Plan plan = dk::backward_plan(N, c->F, d_derr, c->k4_variant, Nf);
    // the direct form adds into the caller's buffer with hardware fp64 atomics (unsafeAtomicAdd): those are only defined on ordinary (coarse-grained)
    // device memory.  Managed / fine-grained memory takes the staged form (partial sums + a reduction launch; on a frame batch: frame by frame, below)
    const int staged_variant = (c->k4_variant < 0 ? 999 : c->k4_variant % 1000) + 1000;
    bool plain = true;
    if (d_grad == grad_xyz && !c->device_args) {
        hipPointerAttribute_t attr;
        plain = hipPointerGetAttributes(&attr, grad_xyz) == hipSuccess && attr.type == hipMemoryTypeDevice;
        if (!plain) (void)hipGetLastError();
    }
    if (plan.direct && !plain) plan = dk::backward_plan(N, c->F, d_derr, staged_variant, Nf);  // Nf < 0 for a batch: frame by frame
    // Round 4, the fused stage (plan.fused): two launches instead of four -- the main pass derives its hypothesis records from the poses in its prologue
    // and (one hypothesis tile: plan.direct) adds the gradient straight into grad_xyz, the finish kernel derives dR/drod itself.  The round-3 staging
    // (k_backward_prep -> main -> k_grad_reduce -> k_support_scatter) remains for the VALU form and behind k4_variant + 1000.
    HIP_TRY(c, c->g6.reserve((size_t)N * 6 * sizeof(double)));
    auto run = [&](int n, const dk::FrameDev& Fd, const dk::K4Plan& pl, size_t h0, size_t cell0, int nf) -> int {  // hypotheses h0 .. h0 + n - 1, gradient rows from cell0
        if (!pl.fused) {
            HIP_TRY(c, c->bwd_staged.reserve(((size_t)n * dk::BWD_STRIDE + (size_t)((n + 15) / 16) * 384) * sizeof(float)));  // records + the K4 LDS image
            HIP_TRY(c, c->dRdH.reserve((size_t)n * dk::BWD_DRDH * sizeof(double)));
        }
        if (!pl.direct) HIP_TRY(c, c->grad_part.reserve((size_t)pl.NT * pl.glayers * P * 3 * sizeof(float)));
        HIP_TRY(c, c->g12_part.reserve((size_t)pl.rows * n * 12 * sizeof(float)));
        const double* ps = d_poses + h0 * 6;
        double* gr = d_grad + cell0 * 3;
        if (!pl.fused) HIP_TRY(c, dk::backward_prep(c->stream, n, ps, Fd, c->bwd_staged.as<float>(), c->dRdH.as<double>()));
        {
            ProfScope ps1(c, 1);
            HIP_TRY(c, dk::score_backward(c->stream, n, c->bwd_staged.as<float>(), Fd, d_derr ? d_derr + h0 * P : nullptr, d_g ? d_g + h0 : nullptr, clampv, tau, beta,
                                          c->grad_part.as<float>(), c->g12_part.as<float>(), pl, ps, gr, flags));
        }
        HIP_TRY(c, dk::score_backward_finish(c->stream, n, Fd, c->grad_part.as<float>(), pl.direct ? 0 : pl.NT * pl.glayers, c->g12_part.as<float>(), pl.rows,
                                             c->dRdH.as<double>(), d_dpnp + h0 * 72, d_sets + h0 * 4, flags, gr, c->g6.as<double>() + h0 * 6,
                                             (pl.variant > 0 && !pl.fused) ? c->bwd_staged.as<float>() : nullptr, pl.fused ? ps : nullptr, nf));
        return DSAC_OK;
    };
    if (plan.Nf < 0) {
        // a frame batch on a map the matrix-core form cannot read as vectors (H*W or W not a multiple of 4, unaligned buffers), or with a hypothesis count
        // per frame that is not a multiple of 16 up to 256: frame by frame through the single-frame forms, the scratch buffers reused in stream order -- F times the launches, the same numbers as F single-frame calls
        for (int f = 0; f < frames; f++) {
            const dk::FrameDev Fd = frame_view(f);
            const dk::K4Plan pf = dk::backward_plan(Nf, Fd, d_derr ? d_derr + (size_t)f * Nf * P : nullptr, plain ? c->k4_variant : staged_variant, 0);
            if (pf.Nf < 0) return fail(c, DSAC_ERR_INVALID, "%s: no kernel form for this map (k4_variant %d)", who, c->k4_variant);
            ARG_TRY(run(Nf, Fd, pf, (size_t)f * Nf, (size_t)f * P, 0));
        }
    } else {
        ARG_TRY(run(N, c->F, plan, 0, 0, Nf));
    }
    c->g6_n = N;
    return end_call(c);
}

int dsac_last_pose_gradients(dsac_ctx* c, int N, double* G6) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_last_pose_gradients: ctx is NULL");
    if (N < 0 || !G6) return fail(c, DSAC_ERR_INVALID, "dsac_last_pose_gradients: NULL argument or negative count");
    if (N > c->g6_n) return fail(c, DSAC_ERR_INVALID, "dsac_last_pose_gradients: %d requested, the last score-backward call had %d", N, c->g6_n);
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    double* d_out;
    ARG_TRY(out_arg(c, G6, (size_t)N * 6, &d_out));
    HIP_TRY(c, hipMemcpyAsync(d_out, c->g6.p, (size_t)N * 6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return end_call(c);
}

int dsac_score_backward(dsac_ctx* c, int N, const double* poses, const int32_t* sets, const float* d_err, const double* dpnp_or_null,
                        unsigned flags, double* grad_xyz) {
    if (c && !d_err) return fail(c, DSAC_ERR_INVALID, "dsac_score_backward: d_err is NULL");
    return score_backward_common(c, "dsac_score_backward", N, poses, sets, d_err, nullptr, 100.0f, 0.f, 0.f, dpnp_or_null, flags, grad_xyz);
}

int dsac_soft_score_backward(dsac_ctx* c, int N, const double* poses, const int32_t* sets, const double* g, float clampv, float tau, float beta,
                             const double* dpnp_or_null, unsigned flags, double* grad_xyz) {
    if (c && !g) return fail(c, DSAC_ERR_INVALID, "dsac_soft_score_backward: g is NULL");
    return score_backward_common(c, "dsac_soft_score_backward", N, poses, sets, nullptr, g, clampv, tau, beta, dpnp_or_null, flags, grad_xyz);
}
int dsac_refine(dsac_ctx* c, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                const int32_t* pert_px_c, const float* pert_value, double* out_poses, int32_t* inlier_map, int32_t* steps_done) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refine: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_refine: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (frames > 1 && (B % frames != 0 || pert_px_c))
        return fail(c, DSAC_ERR_INVALID, "dsac_refine: with a frame batch B must be frames x problems per frame (problem b refines against frame b / (B / frames)), "
                                         "without perturbations");
    if (B < 0 || !init_poses || !perm || !out_poses || steps < 0) return fail(c, DSAC_ERR_INVALID, "dsac_refine: NULL argument or negative count");
    if (max_inl < 1 || max_inl > 256 || min_inl < 0) return fail(c, DSAC_ERR_INVALID, "dsac_refine: need 1 <= max_inl <= 256 (got %d), min_inl >= 0", max_inl);
    if ((pert_px_c != nullptr) != (pert_value != nullptr)) return fail(c, DSAC_ERR_INVALID, "dsac_refine: pert_px_c and pert_value go together");
    if (B == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double* d_init;
    const int32_t *d_perm, *d_px;
    const float* d_pv;
    double* d_out;
    int32_t *d_map, *d_sd;
    ARG_TRY(in_arg(c, init_poses, (size_t)B * 6, &d_init));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, pert_px_c, (size_t)B * 2, &d_px));
    ARG_TRY(in_arg(c, pert_value, (size_t)B, &d_pv));
    ARG_TRY(out_arg(c, out_poses, (size_t)B * 6, &d_out));
    // one frame: the map of problem 0 (H*W counters, += 1 per selection); frame batch: one map per problem (B x H*W)
    ARG_TRY(out_arg(c, inlier_map, frames > 1 ? (size_t)B * P : P, &d_map, /*preload=*/true));
    ARG_TRY(out_arg(c, steps_done, (size_t)B, &d_sd));
    if (dk::refine_split_applies(B, c->F, d_px, nullptr, c->k6_waves)) {  // many problems, long walks: a step as two launches (k_refine.hip)
        HIP_TRY(c, c->k6_scratch.reserve(dk::refine_split_scratch_bytes(B, steps, frames > 1 ? frames : 1, (int)P, max_inl)));
        HIP_TRY(c, dk::refine_split(c->stream, B, d_init, d_perm, steps, max_inl, min_inl, thr, c->F, d_out, d_map, d_sd, frames > 1 ? (int)P : 0,
                                    frames > 1 ? B / frames : 0, c->k6_scratch.p, c->k6_walk_exact));
        return end_call(c);
    }
    HIP_TRY(c, dk::refine(c->stream, B, d_init, d_perm, steps, max_inl, min_inl, thr, d_px, d_pv, c->F, d_out, d_map, d_sd, frames > 1 ? (int)P : 0,
                          frames > 1 ? B / frames : 0, nullptr, nullptr, c->k6_waves));
    return end_call(c);
}

int dsac_refine_fd(dsac_ctx* c, const double* init_pose, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                   const int32_t* inlier_map, float sub_sample, float eps_hyp, float eps_obj, double* J_hyp, int32_t* obj_pixels, double* J_obj,
                   int cap, int32_t* n_obj) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refine_fd: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_refine_fd: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;  // frame batch: every per-image argument holds one slice per frame, one launch per stage
    if (!init_pose || !perm || !inlier_map || !J_hyp || !obj_pixels || !J_obj || !n_obj || cap < 0 || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd: NULL argument or negative count");
    if (max_inl < 1 || max_inl > 256) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd: need 1 <= max_inl <= 256");
    if (!(sub_sample > 0.f) || !(eps_hyp > 0.f) || !(eps_obj > 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd: sub_sample, eps_hyp, eps_obj must be > 0");
    const int skip = (int)(1 / sub_sample);  // core/cnn_softam.h:871
    if (skip < 1) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd: sub_sample > 1");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double* d_init;
    const int32_t *d_perm, *d_map;
    double *d_Jh, *d_Jo;
    int32_t *d_px, *d_n;
    const size_t Fz = (size_t)frames;
    ARG_TRY(in_arg(c, init_pose, Fz * 6, &d_init));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, inlier_map, Fz * P, &d_map));
    ARG_TRY(out_arg(c, J_hyp, Fz * 36, &d_Jh));
    ARG_TRY(out_arg(c, obj_pixels, Fz * (size_t)cap, &d_px));
    ARG_TRY(out_arg(c, J_obj, Fz * (size_t)cap * 18, &d_Jo));
    ARG_TRY(out_arg(c, n_obj, Fz, &d_n));
    const size_t B = (12 + 6 * (size_t)cap) * Fz;
    DevBuf& rp = next_slot(c); HIP_TRY(c, rp.reserve(B * 6 * sizeof(double)));
    DevBuf& rx = next_slot(c); HIP_TRY(c, rx.reserve(B * 2 * sizeof(int32_t)));
    DevBuf& rv = next_slot(c); HIP_TRY(c, rv.reserve(B * sizeof(float)));
    DevBuf& ro = next_slot(c); HIP_TRY(c, ro.reserve(B * 6 * sizeof(double)));
    DevBuf& px = next_slot(c); HIP_TRY(c, px.reserve(((size_t)cap * Fz + 1) * sizeof(int32_t)));
    int32_t* d_pxbuf = d_px ? d_px : px.as<int32_t>();
    int32_t* plan_scratch = nullptr;
    if (const size_t ni = dk::refine_fd_plan_scratch_ints(c->F)) {
        DevBuf& ps = next_slot(c);
        HIP_TRY(c, ps.reserve(ni * Fz * sizeof(int32_t)));
        plan_scratch = ps.as<int32_t>();
    }
    HIP_TRY(c, dk::refine_fd_plan(c->stream, d_init, d_map, c->F, skip, eps_hyp, eps_obj, cap, rp.as<double>(), rx.as<int32_t>(), rv.as<float>(), d_pxbuf, d_n,
                                  plan_scratch, frames, cap));
    HIP_TRY(c, dk::refine_fd_run(c->stream, cap, d_n, rp.as<double>(), d_perm, steps, max_inl, min_inl, thr, rx.as<int32_t>(), rv.as<float>(), c->F, ro.as<double>(),
                                 frames));
    HIP_TRY(c, dk::refine_fd_finish(c->stream, ro.as<double>(), d_n, cap, skip, eps_hyp, eps_obj, d_Jh, d_Jo, frames));
    return end_call(c);
}

int dsac_loss(dsac_ctx* c, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_loss: ctx is NULL");
    if (!est_cv6 || !gt_jp6 || (!out4 && !J6_or_null)) return fail(c, DSAC_ERR_INVALID, "dsac_loss: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const double *d_est, *d_gt;
    double *d_out, *d_J;
    ARG_TRY(in_arg(c, est_cv6, 6, &d_est));
    ARG_TRY(in_arg(c, gt_jp6, 6, &d_gt));
    ARG_TRY(out_arg(c, out4, 4, &d_out));
    ARG_TRY(out_arg(c, J6_or_null, 6, &d_J));
    HIP_TRY(c, dk::pose_loss(c->stream, 1, d_est, d_gt, d_out, d_J));
    return end_call(c);
}

static int refine_fd_sets_common(dsac_ctx* c, const char* who, int M, const int32_t* sets, const int32_t* frame_of_or_null, const int32_t* perm, int steps, int max_inl,
                                 int min_inl, float thr, const int32_t* inlier_maps, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels,
                                 double* J_obj, int cap, int32_t* n_obj) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "%s: ctx is NULL", who);
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "%s: no frame set", who);
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (M < 0 || !sets || !perm || !inlier_maps || !J_set || !obj_pixels || !J_obj || !n_obj || cap < 0 || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "%s: NULL argument or negative count", who);
    if (frames > 1 && !frame_of_or_null && M % frames != 0)
        return fail(c, DSAC_ERR_INVALID, "%s: with a frame batch M must be frames x (hypotheses per frame) -- or give every hypothesis its frame (dsac_refine_fd_sets_frames)", who);
    if (max_inl < 1 || max_inl > 256) return fail(c, DSAC_ERR_INVALID, "%s: need 1 <= max_inl <= 256", who);
    if (!(sub_sample > 0.f) || !(eps_obj > 0.f)) return fail(c, DSAC_ERR_INVALID, "%s: sub_sample and eps_obj must be > 0", who);
    const int skip = (int)(1 / sub_sample);  // core/cnn.h:933
    if (skip < 1) return fail(c, DSAC_ERR_INVALID, "%s: sub_sample > 1", who);
    if (M == 0) return DSAC_OK;
    const size_t R = 18 + 6 * (size_t)cap, B = R * (size_t)M;
    if (B > (1u << 26)) return fail(c, DSAC_ERR_INVALID, "%s: M * (18 + 6*cap) = %zu replicas is too many", who, B);
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const int32_t *d_sets, *d_perm, *d_maps, *d_fof = nullptr;
    double *d_Js, *d_Jo;
    int32_t *d_px, *d_n;
    ARG_TRY(in_arg(c, sets, (size_t)M * 4, &d_sets));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, inlier_maps, (size_t)M * P, &d_maps));
    std::vector<int32_t> implicit_frames;  // pageable host memory: the upload below has left it when hipMemcpyAsync returns
    if (frames > 1) {
        if (frame_of_or_null) {
            if (!is_device_ptr(frame_of_or_null, c))
                for (int m = 0; m < M; m++)
                    if (frame_of_or_null[m] < 0 || frame_of_or_null[m] >= frames) return fail(c, DSAC_ERR_INVALID, "%s: frame_of[%d] = %d is not a frame of the batch", who, m, frame_of_or_null[m]);
            ARG_TRY(in_arg(c, frame_of_or_null, (size_t)M, &d_fof));
        } else {
            implicit_frames.resize(M);
            for (int m = 0; m < M; m++) implicit_frames[m] = m / (M / frames);
            DevBuf& fb = next_slot(c);
            HIP_TRY(c, fb.reserve((size_t)M * sizeof(int32_t)));
            HIP_TRY(c, hipMemcpyAsync(fb.p, implicit_frames.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));  // the vector goes out of scope with this call
            d_fof = fb.as<int32_t>();
        }
    }
    ARG_TRY(out_arg(c, J_set, (size_t)M * 54, &d_Js));
    ARG_TRY(out_arg(c, obj_pixels, (size_t)M * cap, &d_px));
    ARG_TRY(out_arg(c, J_obj, (size_t)M * cap * 18, &d_Jo));
    ARG_TRY(out_arg(c, n_obj, (size_t)M, &d_n));
    DevBuf& rp = next_slot(c); HIP_TRY(c, rp.reserve(B * 6 * sizeof(double)));
    DevBuf& rx = next_slot(c); HIP_TRY(c, rx.reserve(B * 2 * sizeof(int32_t)));
    DevBuf& rv = next_slot(c); HIP_TRY(c, rv.reserve(B * sizeof(float)));
    DevBuf& ro = next_slot(c); HIP_TRY(c, ro.reserve(B * 6 * sizeof(double)));
    int32_t* plan_scratch = nullptr;
    if (const size_t ni = dk::refine_fd_plan_scratch_ints(c->F)) {
        DevBuf& ps = next_slot(c);
        HIP_TRY(c, ps.reserve(ni * (size_t)M * sizeof(int32_t)));
        plan_scratch = ps.as<int32_t>();
    }
    HIP_TRY(c, dk::refine_fd_plan_set(c->stream, d_sets, d_maps, c->F, skip, eps_obj, cap, rp.as<double>(), rx.as<int32_t>(), rv.as<float>(), d_px, d_n, M,
                                      plan_scratch, d_fof));
    HIP_TRY(c, dk::refine_fd_run_set(c->stream, cap, d_n, rp.as<double>(), d_perm, steps, max_inl, min_inl, thr, rx.as<int32_t>(), rv.as<float>(), c->F,
                                     ro.as<double>(), M, d_fof));
    HIP_TRY(c, dk::refine_fd_finish_set(c->stream, ro.as<double>(), d_n, cap, skip, eps_obj, d_Js, d_Jo, M));
    return end_call(c);
}

int dsac_refine_fd_sets(dsac_ctx* c, int M, const int32_t* sets, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                        const int32_t* inlier_maps, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj, int cap,
                        int32_t* n_obj) {
    return refine_fd_sets_common(c, "dsac_refine_fd_sets", M, sets, nullptr, perm, steps, max_inl, min_inl, thr, inlier_maps, sub_sample, eps_obj, J_set, obj_pixels, J_obj,
                                 cap, n_obj);
}

int dsac_refine_fd_sets_frames(dsac_ctx* c, int M, const int32_t* sets, const int32_t* frame_of, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                               const int32_t* inlier_maps, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj, int cap,
                               int32_t* n_obj) {
    if (c && !frame_of) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_sets_frames: frame_of is NULL");
    return refine_fd_sets_common(c, "dsac_refine_fd_sets_frames", M, sets, frame_of, perm, steps, max_inl, min_inl, thr, inlier_maps, sub_sample, eps_obj, J_set, obj_pixels,
                                 J_obj, cap, n_obj);
}

int dsac_gather_patches(dsac_ctx* c, const uint8_t* bgr, int H, int W, const int32_t* sampling_xy, int n, int patch, float* patches, int32_t* skipped_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_gather_patches: ctx is NULL");
    if (!bgr || !sampling_xy || !patches || H <= 0 || W <= 0 || n < 0 || patch <= 0 || (patch & 1))
        return fail(c, DSAC_ERR_INVALID, "dsac_gather_patches: NULL argument, empty image or odd/non-positive patch size");
    if (n == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const uint8_t* d_img;
    const int32_t* d_xy;
    float* d_out;
    int32_t* d_skip;
    ARG_TRY(in_arg(c, bgr, (size_t)H * W * 3, &d_img));
    ARG_TRY(in_arg(c, sampling_xy, (size_t)n * 2, &d_xy));
    ARG_TRY(out_arg(c, patches, (size_t)n * 3 * patch * patch, &d_out));
    ARG_TRY(out_arg(c, skipped_or_null, 1, &d_skip));
    if (d_skip) HIP_TRY(c, hipMemsetAsync(d_skip, 0, sizeof(int32_t), c->stream));
    HIP_TRY(c, dk::gather_patches(c->stream, d_img, H, W, d_xy, n, patch, d_out, d_skip));
    return end_call(c);
}

int dsac_loss_batch(dsac_ctx* c, int B, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_loss_batch: ctx is NULL");
    if (B < 0 || !est_cv6 || !gt_jp6 || (!out4 && !J6_or_null)) return fail(c, DSAC_ERR_INVALID, "dsac_loss_batch: NULL argument or negative count");
    if (B == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const double *d_est, *d_gt;
    double *d_out, *d_J;
    ARG_TRY(in_arg(c, est_cv6, (size_t)B * 6, &d_est));
    ARG_TRY(in_arg(c, gt_jp6, 6, &d_gt));
    ARG_TRY(out_arg(c, out4, (size_t)B * 4, &d_out));
    ARG_TRY(out_arg(c, J6_or_null, (size_t)B * 6, &d_J));
    HIP_TRY(c, dk::pose_loss(c->stream, B, d_est, d_gt, d_out, d_J));
    return end_call(c);
}

int dsac_refine_all(dsac_ctx* c, int N, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                    const int32_t* sets_or_null, double* out_poses, int32_t* inlier_maps_or_null, int32_t* steps_done_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refine_all: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_refine_all: no frame set");
    // frame batch: N = frames x hypotheses per frame, hypothesis h refines against frame h / (N / frames) -- frames x N refinement problems in ONE launch
    // (core/cnn.h:1155-1215 is the loop over the hypotheses of one image; 16 images x 256 hypotheses are 4096 waves, which fill the chip where 256 do not)
    const int frames_ra = c->F.frames > 1 ? c->F.frames : 1;
    if (frames_ra > 1 && N % frames_ra != 0) return fail(c, DSAC_ERR_INVALID, "dsac_refine_all: with a frame batch N must be frames x (hypotheses per frame), got %d for %d frames", N, frames_ra);
    if (N < 0 || !init_poses || !perm || !out_poses || steps < 0) return fail(c, DSAC_ERR_INVALID, "dsac_refine_all: NULL argument or negative count");
    if (max_inl < 1 || max_inl > 256 || min_inl < 0) return fail(c, DSAC_ERR_INVALID, "dsac_refine_all: need 1 <= max_inl <= 256 (got %d), min_inl >= 0", max_inl);
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double* d_init;
    const int32_t *d_perm, *d_sets;
    double* d_out;
    int32_t *d_maps, *d_sd;
    ARG_TRY(in_arg(c, init_poses, (size_t)N * 6, &d_init));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, sets_or_null, (size_t)N * 4, &d_sets));
    ARG_TRY(out_arg(c, out_poses, (size_t)N * 6, &d_out));
    ARG_TRY(out_arg(c, inlier_maps_or_null, (size_t)N * P, &d_maps, /*preload=*/false));
    ARG_TRY(out_arg(c, steps_done_or_null, (size_t)N, &d_sd));
    if (d_maps) HIP_TRY(c, hipMemsetAsync(d_maps, 0, (size_t)N * P * sizeof(int32_t), c->stream));
    if (dk::refine_split_applies(N, c->F, nullptr, nullptr, c->k6_waves)) {  // the DSAC variant on a big map: every hypothesis walks most of it -- walk and LM as separate launches
        HIP_TRY(c, c->k6_scratch.reserve(dk::refine_split_scratch_bytes(N, steps, frames_ra > 1 ? frames_ra : 1, (int)P, max_inl)));
        HIP_TRY(c, dk::refine_split(c->stream, N, d_init, d_perm, steps, max_inl, min_inl, thr, c->F, d_out, d_maps, d_sd, d_maps ? (int)P : 0,
                                    frames_ra > 1 ? N / frames_ra : 0, c->k6_scratch.p, c->k6_walk_exact));
    } else
    HIP_TRY(c, dk::refine(c->stream, N, d_init, d_perm, steps, max_inl, min_inl, thr, nullptr, nullptr, c->F, d_out, d_maps, d_sd, d_maps ? (int)P : 0,
                          frames_ra > 1 ? N / frames_ra : 0, nullptr, nullptr, c->k6_waves));
    if (d_maps && d_sets) HIP_TRY(c, dk::zero_set_cells(c->stream, N, d_sets, (int)P, d_maps));
    return end_call(c);
}

int dsac_refine_fd_set(dsac_ctx* c, const int32_t* set4, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                       const int32_t* inlier_map, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj, int cap,
                       int32_t* n_obj) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_refine_fd_set: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_refine_fd_set: no frame set");
    if (c->F.frames > 1) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_set: a frame batch is set (one hypothesis: dsac_set_frame of its image, or dsac_refine_fd_sets_frames)");
    if (!set4 || !perm || !inlier_map || !J_set || !obj_pixels || !J_obj || !n_obj || cap < 0 || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_set: NULL argument or negative count");
    if (max_inl < 1 || max_inl > 256) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_set: need 1 <= max_inl <= 256");
    if (!(sub_sample > 0.f) || !(eps_obj > 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_set: sub_sample and eps_obj must be > 0");
    const int skip = (int)(1 / sub_sample);  // core/cnn.h:933
    if (skip < 1) return fail(c, DSAC_ERR_INVALID, "dsac_refine_fd_set: sub_sample > 1");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const int32_t *d_set, *d_perm, *d_map;
    double *d_Js, *d_Jo;
    int32_t *d_px, *d_n;
    ARG_TRY(in_arg(c, set4, 4, &d_set));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, inlier_map, P, &d_map));
    ARG_TRY(out_arg(c, J_set, 54, &d_Js));
    ARG_TRY(out_arg(c, obj_pixels, (size_t)cap, &d_px));
    ARG_TRY(out_arg(c, J_obj, (size_t)cap * 18, &d_Jo));
    ARG_TRY(out_arg(c, n_obj, 1, &d_n));
    const size_t B = 18 + 6 * (size_t)cap;
    DevBuf& rp = next_slot(c); HIP_TRY(c, rp.reserve(B * 6 * sizeof(double)));
    DevBuf& rx = next_slot(c); HIP_TRY(c, rx.reserve(B * 2 * sizeof(int32_t)));
    DevBuf& rv = next_slot(c); HIP_TRY(c, rv.reserve(B * sizeof(float)));
    DevBuf& ro = next_slot(c); HIP_TRY(c, ro.reserve(B * 6 * sizeof(double)));
    DevBuf& px = next_slot(c); HIP_TRY(c, px.reserve(((size_t)cap + 1) * sizeof(int32_t)));
    int32_t* d_pxbuf = d_px ? d_px : px.as<int32_t>();
    int32_t* plan_scratch = nullptr;
    if (const size_t ni = dk::refine_fd_plan_scratch_ints(c->F)) {
        DevBuf& ps = next_slot(c);
        HIP_TRY(c, ps.reserve(ni * sizeof(int32_t)));
        plan_scratch = ps.as<int32_t>();
    }
    HIP_TRY(c, dk::refine_fd_plan_set(c->stream, d_set, d_map, c->F, skip, eps_obj, cap, rp.as<double>(), rx.as<int32_t>(), rv.as<float>(), d_pxbuf, d_n, 1,
                                      plan_scratch));
    HIP_TRY(c, dk::refine_fd_run_set(c->stream, cap, d_n, rp.as<double>(), d_perm, steps, max_inl, min_inl, thr, rx.as<int32_t>(), rv.as<float>(), c->F, ro.as<double>()));
    HIP_TRY(c, dk::refine_fd_finish_set(c->stream, ro.as<double>(), d_n, cap, skip, eps_obj, d_Js, d_Jo));
    return end_call(c);
}

int dsac_backward_path1(dsac_ctx* c, int N, const double* poses, const int32_t* sets, const double* w, const double* avg_cv6, const double* ref_cv6,
                        const double* gt_jp6, const int32_t* perm, int steps, int max_inl, int min_inl, float thr, const int32_t* inlier_map, float sub_sample,
                        float eps_hyp, float eps_obj, double g_scale, double* dpnp_out_or_null, double* grad_xyz, double* g, double* dL_out_or_null,
                        double* v6_out_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_backward_path1: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_backward_path1: no frame set");
    // frame batch: N = frames x hypotheses per frame; every per-hypothesis array is frame-major, every per-image argument (avg, ref, gt, inlier map, grad_xyz,
    // dL, v6) holds one slice per frame; each stage of the chain is ONE launch over all frames (core/train_ransac_softam.cpp:288-376 is per image)
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (N <= 0 || N % frames != 0) return fail(c, DSAC_ERR_INVALID, "dsac_backward_path1: N must be positive and frames x (hypotheses per frame)");
    if (!poses || !sets || !w || !avg_cv6 || !ref_cv6 || !gt_jp6 || !perm || !inlier_map || !grad_xyz || !g || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_backward_path1: NULL argument or bad count");
    if (max_inl < 1 || max_inl > 256) return fail(c, DSAC_ERR_INVALID, "dsac_backward_path1: need 1 <= max_inl <= 256");
    if (!(sub_sample > 0.f) || !(eps_hyp > 0.f) || !(eps_obj > 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_backward_path1: sub_sample, eps_hyp, eps_obj must be > 0");
    const int skip = (int)(1 / sub_sample);  // core/cnn_softam.h:871
    if (skip < 1) return fail(c, DSAC_ERR_INVALID, "dsac_backward_path1: sub_sample > 1");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P, Fz = (size_t)frames;
    const int Nf = N / frames;
    const double *d_poses, *d_w, *d_avg, *d_ref, *d_gt;
    const int32_t *d_sets, *d_perm, *d_map;
    double *d_dpnp, *d_grad, *d_g, *d_dL, *d_v6;
    ARG_TRY(in_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(in_arg(c, sets, (size_t)N * 4, &d_sets));
    ARG_TRY(in_arg(c, w, (size_t)N, &d_w));
    ARG_TRY(in_arg(c, avg_cv6, Fz * 6, &d_avg));
    ARG_TRY(in_arg(c, ref_cv6, Fz * 6, &d_ref));
    ARG_TRY(in_arg(c, gt_jp6, Fz * 6, &d_gt));
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, inlier_map, Fz * P, &d_map));
    ARG_TRY(out_arg(c, dpnp_out_or_null, (size_t)N * 72, &d_dpnp));
    ARG_TRY(out_arg(c, grad_xyz, Fz * P * 3, &d_grad, /*preload=*/true));
    ARG_TRY(out_arg(c, g, (size_t)N, &d_g));
    ARG_TRY(out_arg(c, dL_out_or_null, Fz * 6, &d_dL));
    ARG_TRY(out_arg(c, v6_out_or_null, Fz * 6, &d_v6));
    // the inlier map holds at most steps * max_inl hits, every skip-th of them is differentiated
    const int cap = (int)std::min<size_t>(4096, (size_t)steps * (size_t)max_inl / (size_t)skip + 1);
    const size_t R = 12 + 6 * (size_t)cap, B = R * Fz;
    DevBuf& sc = next_slot(c);  // per frame: dL 6 | out4 4 | v6 6 | J_hyp 36 | J_obj cap*18 ; then rep poses B*6 | rep out B*6   (doubles)
    const size_t nd = Fz * (6 + 4 + 6 + 36 + (size_t)cap * 18) + B * 6 + B * 6;
    HIP_TRY(c, sc.reserve(nd * sizeof(double)));
    double* base = sc.as<double>();
    double *s_dL = d_dL ? d_dL : base, *s_out4 = base + Fz * 6, *s_v6 = d_v6 ? d_v6 : base + Fz * 10, *s_Jh = base + Fz * 16, *s_Jo = base + Fz * 52;
    double *s_rp = s_Jo + Fz * (size_t)cap * 18, *s_ro = s_rp + B * 6;
    DevBuf& si = next_slot(c);  // rep px/c B*2 | obj pixels frames*cap + 1 | n frames  (int32) ; rep value B (float)
    HIP_TRY(c, si.reserve((B * 2 + Fz * (size_t)cap + 1 + Fz) * sizeof(int32_t) + B * sizeof(float)));
    int32_t* s_rx = si.as<int32_t>();
    int32_t* s_px = s_rx + B * 2;
    int32_t* s_n = s_px + Fz * (size_t)cap + 1;
    float* s_rv = reinterpret_cast<float*>(s_n + Fz);
    if (!d_dpnp) {
        DevBuf& sd = next_slot(c);
        HIP_TRY(c, sd.reserve((size_t)N * 72 * sizeof(double)));
        d_dpnp = sd.as<double>();
    }
    // K5 on a side stream: dPNP (:344-349) depends on nothing but the minimal sets, and the chain below contains K6's finite-difference replicas -- one
    // LM chain long (~100 us) with most of the chip idle: the 24 x N P3P solves run beside it (round 5: the 40 x 40, 16-frame training round 381 -> 364 us).
    // Forked behind everything enqueued so far (the staged arguments, earlier readers of the dPNP buffer), joined before its first reader.
    if (!c->bwd_side) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->bwd_side, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->bwd_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->bwd_join, hipEventDisableTiming));
    }
    // dLossMax at the refined pose (train_ransac_softam.cpp:301-304), one ground truth per frame
    HIP_TRY(c, dk::pose_loss(c->stream, frames, d_ref, d_gt, s_out4, s_dL, 6));
    // dRefineObj / dRefineHyp as one batch of finite-difference replicas (:307-341)
    int32_t* plan_scratch = nullptr;
    if (const size_t ni = dk::refine_fd_plan_scratch_ints(c->F)) {
        DevBuf& ps = next_slot(c);
        HIP_TRY(c, ps.reserve(ni * Fz * sizeof(int32_t)));
        plan_scratch = ps.as<int32_t>();
    }
    HIP_TRY(c, dk::refine_fd_plan(c->stream, d_avg, d_map, c->F, skip, eps_hyp, eps_obj, cap, s_rp, s_rx, s_rv, s_px, s_n, plan_scratch, frames, cap));
    // the two cross-stream hand-overs cost ~7 us each (profiles/r05_k5_side_stream.txt): worth it from about 3 000 minimal sets (K5: 10 ns per set)
    const bool k5_beside = N >= 3072;
    if (k5_beside) HIP_TRY(c, hipEventRecord(c->bwd_fork, c->stream));  // the fork point is the end of the plan: K5 becomes eligible together with the replicas ...
    HIP_TRY(c, dk::refine_fd_run(c->stream, cap, s_n, s_rp, d_perm, steps, max_inl, min_inl, thr, s_rx, s_rv, c->F, s_ro, frames));
    // ... but is enqueued behind them: the replicas' few hundred waves take their SIMDs first, K5's thousands fill in around them (launched in front,
    // K5 held every wave slot for its first 20 us and the chain started late: 127 against 101 us, nothing gained)
    if (k5_beside) {
        HIP_TRY(c, hipStreamWaitEvent(c->bwd_side, c->bwd_fork, 0));
        HIP_TRY(c, dk::dpnp(c->bwd_side, N, d_sets, c->F, 0.1f, d_dpnp, Nf));
        HIP_TRY(c, hipEventRecord(c->bwd_join, c->bwd_side));
    }
    HIP_TRY(c, dk::refine_fd_finish(c->stream, s_ro, s_n, cap, skip, eps_hyp, eps_obj, s_Jh, s_Jo, frames));
    HIP_TRY(c, dk::path1_assemble(c->stream, s_dL, s_Jh, s_px, s_Jo, s_n, cap, (int)P, d_grad, s_v6, frames, cap));
    // sum_h w_h dPNP_h to the support points and the softmax backward (:344-376)
    if (k5_beside) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->bwd_join, 0));
    else HIP_TRY(c, dk::dpnp(c->stream, N, d_sets, c->F, 0.1f, d_dpnp, Nf));
    HIP_TRY(c, dk::path1_softmax_backward(c->stream, Nf, c->F.P, s_v6, d_w, d_poses, d_sets, d_dpnp, d_grad, d_g, g_scale, frames));
    return end_call(c);
}

int dsac_select(dsac_ctx* c, int N, const double* probs, const double* losses, int loss_stride, double u, int32_t* hyp_idx_or_null, double* expected_loss_or_null,
                double* score_gradients_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_select: ctx is NULL");
    if (N <= 0 || !probs || !losses || loss_stride < 1 || !(u < 1.0)) return fail(c, DSAC_ERR_INVALID, "dsac_select: need N > 0, probs, losses, loss_stride >= 1 and u < 1");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const double *d_w, *d_l;
    int32_t* d_idx;
    double *d_e, *d_g;
    ARG_TRY(in_arg(c, probs, (size_t)N, &d_w));
    ARG_TRY(in_arg(c, losses, (size_t)N * loss_stride, &d_l));
    ARG_TRY(out_arg(c, hyp_idx_or_null, 1, &d_idx));
    ARG_TRY(out_arg(c, expected_loss_or_null, 1, &d_e));
    ARG_TRY(out_arg(c, score_gradients_or_null, (size_t)N, &d_g));
    HIP_TRY(c, dk::dsac_select(c->stream, N, d_w, d_l, loss_stride, u, 1e-8, d_idx, d_e, d_g));
    return end_call(c);
}

int dsac_loss_frames(dsac_ctx* c, int B, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_loss_frames: ctx is NULL");
    if (B < 0 || !est_cv6 || !gt_jp6 || (!out4 && !J6_or_null)) return fail(c, DSAC_ERR_INVALID, "dsac_loss_frames: NULL argument or negative count");
    if (B == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const double *d_est, *d_gt;
    double *d_out, *d_J;
    ARG_TRY(in_arg(c, est_cv6, (size_t)B * 6, &d_est));
    ARG_TRY(in_arg(c, gt_jp6, (size_t)B * 6, &d_gt));
    ARG_TRY(out_arg(c, out4, (size_t)B * 4, &d_out));
    ARG_TRY(out_arg(c, J6_or_null, (size_t)B * 6, &d_J));
    HIP_TRY(c, dk::pose_loss(c->stream, B, d_est, d_gt, d_out, d_J, 6));
    return end_call(c);
}

int dsac_loss_batch_frames(dsac_ctx* c, int frames, int per_frame, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_loss_batch_frames: ctx is NULL");
    if (frames < 0 || per_frame <= 0 || !est_cv6 || !gt_jp6 || (!out4 && !J6_or_null)) return fail(c, DSAC_ERR_INVALID, "dsac_loss_batch_frames: NULL argument or bad count");
    if ((long long)frames * per_frame > (1ll << 26)) return fail(c, DSAC_ERR_INVALID, "dsac_loss_batch_frames: too many estimates");
    if (frames == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t B = (size_t)frames * per_frame;
    const double *d_est, *d_gt;
    double *d_out, *d_J;
    ARG_TRY(in_arg(c, est_cv6, B * 6, &d_est));
    ARG_TRY(in_arg(c, gt_jp6, (size_t)frames * 6, &d_gt));
    ARG_TRY(out_arg(c, out4, B * 4, &d_out));
    ARG_TRY(out_arg(c, J6_or_null, B * 6, &d_J));
    HIP_TRY(c, dk::pose_loss(c->stream, (int)B, d_est, d_gt, d_out, d_J, 6, per_frame));
    return end_call(c);
}

int dsac_select_frames(dsac_ctx* c, int frames, int N, const double* probs, const double* losses, int loss_stride, const double* u, int32_t* hyp_idx_or_null,
                       double* expected_loss_or_null, double* score_gradients_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_select_frames: ctx is NULL");
    if (frames <= 0 || N <= 0 || !probs || !losses || !u || loss_stride < 1) return fail(c, DSAC_ERR_INVALID, "dsac_select_frames: need frames > 0, N > 0, probs, losses, u, loss_stride >= 1");
    if ((long long)frames * N > (1ll << 24)) return fail(c, DSAC_ERR_INVALID, "dsac_select_frames: too many hypotheses");
    if (!is_device_ptr(u, c))
        for (int f = 0; f < frames; f++)
            if (!(u[f] < 1.0)) return fail(c, DSAC_ERR_INVALID, "dsac_select_frames: u[%d] must be < 1 (negative: the most probable entry)", f);
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t NT = (size_t)frames * N;
    const double *d_w, *d_l, *d_u;
    int32_t* d_idx;
    double *d_e, *d_g;
    ARG_TRY(in_arg(c, probs, NT, &d_w));
    ARG_TRY(in_arg(c, losses, NT * loss_stride, &d_l));
    ARG_TRY(in_arg(c, u, (size_t)frames, &d_u));
    ARG_TRY(out_arg(c, hyp_idx_or_null, (size_t)frames, &d_idx));
    ARG_TRY(out_arg(c, expected_loss_or_null, (size_t)frames, &d_e));
    ARG_TRY(out_arg(c, score_gradients_or_null, NT, &d_g));
    HIP_TRY(c, dk::dsac_select_frames(c->stream, frames, N, d_w, d_l, loss_stride, d_u, 1e-8, d_idx, d_e, d_g));
    return end_call(c);
}

// d_err[h][p] = g[h] * d soft[h] / d err[h][p] = g[h] * (-beta) * s (1 - s),  s = sigmoid(beta (tau - err[h][p])); zero where the residual sits on the clamp
// (the score no longer depends on it there -- what K4's in-kernel form does, k_backward.hip).  One float4 per lane, the row index from the launch's y.
namespace {
__global__ __launch_bounds__(256) void k_soft_derr(int P4, const float4* __restrict__ err, const double* __restrict__ g, float clampv, float tau, float beta,
                                                   float4* __restrict__ d_err) {
    const size_t h = blockIdx.y;
    const float gh = (float)g[h] * (-beta);
    const float4* e = err + h * (size_t)P4;
    float4* d = d_err + h * (size_t)P4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P4; i += gridDim.x * blockDim.x) {
        const float4 v = e[i];
        float4 o;
        const float in[4] = {v.x, v.y, v.z, v.w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float s = 1.f / (1.f + __expf(-beta * (tau - in[k])));
            out[k] = in[k] >= clampv ? 0.f : gh * s * (1.f - s);
        }
        o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
        d[i] = o;
    }
}
}  // namespace

int dsac_soft_score_derr(dsac_ctx* c, int N, const double* g, const float* err, float clampv, float tau, float beta, float* d_err) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_soft_score_derr: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_soft_score_derr: no frame set (the maps are H*W wide)");
    if (N < 0 || !g || !err || !d_err || !(beta > 0.f)) return fail(c, DSAC_ERR_INVALID, "dsac_soft_score_derr: NULL argument, negative count or beta <= 0");
    if (c->F.P % 4 != 0) return fail(c, DSAC_ERR_INVALID, "dsac_soft_score_derr: H*W must be a multiple of 4");
    if (N == 0) return DSAC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double* d_g;
    const float* d_e;
    float* d_o;
    ARG_TRY(in_arg(c, g, (size_t)N, &d_g));
    ARG_TRY(in_arg(c, err, (size_t)N * P, &d_e));
    ARG_TRY(out_arg(c, d_err, (size_t)N * P, &d_o));
    if ((reinterpret_cast<uintptr_t>(d_e) | reinterpret_cast<uintptr_t>(d_o)) % 16 != 0) return fail(c, DSAC_ERR_INVALID, "dsac_soft_score_derr: err / d_err must be 16-byte aligned");
    const int P4 = (int)(P / 4);
    const unsigned gx = (unsigned)std::min<int>((P4 + 255) / 256, 64);
    for (int h0 = 0; h0 < N; h0 += 65535) {
        const int n = std::min(N - h0, 65535);
        hipLaunchKernelGGL(k_soft_derr, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, P4, reinterpret_cast<const float4*>(d_e + (size_t)h0 * P), d_g + h0, clampv, tau, beta,
                           reinterpret_cast<float4*>(d_o + (size_t)h0 * P));
        HIP_TRY(c, hipGetLastError());
    }
    return end_call(c);
}

// ---- shared by dsac_process_images and the begin / finish pair -------------------------------------------------------------------------------
// Which deferral mode a call runs in: the context's "pi_defer_tail", but only with device-resident arguments -- a host destination is copied back at
// the end of the call, and a host `perm` / `gt` lives in a staging slot that the next call reuses.
static int pi_mode(dsac_ctx* c, const void* perm, const void* gt_or_null) {
    return (c->pending.empty() && is_device_ptr(perm, c) && (!gt_or_null || is_device_ptr(gt_or_null, c))) ? c->pi_defer_tail : 0;
}
// mode 2, which tail stream: consecutive calls write different arrays there, so their tails are independent -- SMALL calls (up to two full-size
// images' worth of hypothesis x cell pairs: K1 + K2 shorter than the one-wave refinement chain) alternate between two streams and two tails run
// side by side; larger calls hide their tail under the next call anyway and stay on the first stream (measured: with both streams in use for
// 8-image calls configs[3]'s rank step went from 0.51 to 0.64 ms in the bench process -- the tails then ran exposed; the cause was not isolated,
// GPU_MAX_HW_QUEUES = 8 changes neither number, profiles/r04_hw_queues.txt)
static int pi_tail_index(int mode, int b, long long N, long long P) { return (mode == 2 && N * P <= 2ll * 256 * 307200) ? b : 0; }
// streams and events of the deferred tails, created on first use
static int pi_tail_setup(dsac_ctx* c, int mode, int tk) {
    if (mode == 0) return DSAC_OK;
    if (!c->tail_go) HIP_TRY(c, hipEventCreateWithFlags(&c->tail_go, hipEventDisableTiming));
    if (!c->tail[tk]) {
        // the tail's launches are one to a few waves each and latency-bound: with the highest stream priority their workgroups are placed ahead of the
        // thousands K2 still has pending ("tail_prio" 0: default priority, for the A/B)
        int lo = 0, hi = 0;
        if (c->tail_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
            HIP_TRY(c, hipStreamCreateWithPriority(&c->tail[tk], hipStreamNonBlocking, hi));
        else
            HIP_TRY(c, hipStreamCreateWithFlags(&c->tail[tk], hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->tail_done[tk], hipEventDisableTiming));
    }
    if (mode == 2 && !c->pi_k2done) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->pi_k2done, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) HIP_TRY(c, hipEventCreateWithFlags(&c->pi_scored[k], hipEventDisableTiming));
    }
    return DSAC_OK;
}
// K6 (+ K7 at the end of its wave) of every frame on `ts`; with a deferred tail its completion event is recorded and the tail marked pending
static int pi_refine_tail(dsac_ctx* c, hipStream_t ts, bool defer, int tk, int frames, const double* d_avg, const int32_t* d_perm, int steps, int max_inl,
                          int min_inl, float thr, double* d_ref, int32_t* d_maps, int32_t* d_sd, const double* d_gt, double* d_out4) {
    const size_t P = (size_t)c->F.P;
    if (d_maps) HIP_TRY(c, hipMemsetAsync(d_maps, 0, (size_t)frames * P * sizeof(int32_t), ts));
    // K7 rides at the end of K6's wave: the loss of a refined pose is computed by the lane that holds it (one launch less behind the refinement chain)
    HIP_TRY(c, dk::refine(ts, frames, d_avg, d_perm, steps, max_inl, min_inl, (float)(int)thr, nullptr, nullptr, c->F, d_ref, d_maps, d_sd,
                          d_maps ? (int)P : 0, frames > 1 ? 1 : 0, d_out4 ? d_gt : nullptr, d_out4, c->k6_waves));
    if (defer) {
        HIP_TRY(c, hipEventRecord(c->tail_done[tk], ts));
        c->tail_pending[tk] = true;
    }
    return DSAC_OK;
}

int dsac_process_images(dsac_ctx* c, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clampv, float tau, float beta, double scale,
                        const int32_t* perm, int steps, int max_inl, int min_inl, const double* gt_jp6_or_null, double* poses, int32_t* sets_out, uint8_t* ok,
                        float* err_or_null, double* scores_or_null, double* w, double* entropy, double* avg6, double* ref6, int32_t* steps_done,
                        int32_t* inlier_maps_or_null, double* out4_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_process_images: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_process_images: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (hyps_per_frame <= 0 || (frames > 1 && hyps_per_frame % dk::K2_NF_MULTIPLE != 0))
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images: hyps_per_frame must be positive (a multiple of %d for a frame batch), got %d", dk::K2_NF_MULTIPLE,
                    hyps_per_frame);
    if ((long long)hyps_per_frame * frames > (1ll << 24)) return fail(c, DSAC_ERR_INVALID, "dsac_process_images: too many hypotheses");
    if (!perm || !poses || !sets_out || !ok || !w || !entropy || !avg6 || !ref6 || !steps_done || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images: NULL argument or negative step count");
    if ((out4_or_null != nullptr) != (gt_jp6_or_null != nullptr)) return fail(c, DSAC_ERR_INVALID, "dsac_process_images: out4 and gt_jp6 go together");
    if (max_inl < 1 || max_inl > 256 || min_inl < 0) return fail(c, DSAC_ERR_INVALID, "dsac_process_images: need 1 <= max_inl <= 256 (got %d), min_inl >= 0", max_inl);
    if (max_tries <= 0 || c->F.P < 4) return fail(c, DSAC_ERR_INVALID, "dsac_process_images: max_tries > 0 and a frame of at least 4 cells needed");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c, /*keep_tail=*/c->pi_defer_tail != 0);
    const size_t P = (size_t)c->F.P;
    const int N = hyps_per_frame * frames, Nf = frames > 1 ? hyps_per_frame : 0;
    const int32_t* d_perm;
    const double* d_gt;
    double *d_poses, *d_scores, *d_w, *d_ent, *d_avg, *d_ref, *d_out4;
    int32_t *d_sets, *d_sd, *d_maps;
    uint8_t* d_ok;
    float* d_err;
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, gt_jp6_or_null, (size_t)frames * 6, &d_gt));
    ARG_TRY(out_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, sets_out, (size_t)N * 4, &d_sets));
    ARG_TRY(out_arg(c, ok, (size_t)N, &d_ok));
    ARG_TRY(out_arg(c, err_or_null, (size_t)N * P, &d_err));
    ARG_TRY(out_arg(c, scores_or_null, (size_t)N, &d_scores));
    ARG_TRY(out_arg(c, w, (size_t)N, &d_w));
    ARG_TRY(out_arg(c, entropy, (size_t)frames, &d_ent));
    ARG_TRY(out_arg(c, avg6, (size_t)frames * 6, &d_avg));
    ARG_TRY(out_arg(c, ref6, (size_t)frames * 6, &d_ref));
    ARG_TRY(out_arg(c, steps_done, (size_t)frames, &d_sd));
    ARG_TRY(out_arg(c, inlier_maps_or_null, (size_t)frames * P, &d_maps));
    ARG_TRY(out_arg(c, out4_or_null, (size_t)frames * 4, &d_out4));
    // "pi_defer_tail" = 1: the refinement tail (K6, a 90 us latency chain on one wave per frame, and K7) goes to its own stream and runs under K1 / K2 of
    // the NEXT dsac_process_images call.  Its outputs (ref6, steps_done, inlier maps, out4) are complete in the order of the context's stream only after
    // the next call of any other entry point, dsac_join_tail or dsac_synchronize.  = 2: the score tail (reduction of the per-tile sums, K3) goes there
    // too -- K1 of the next call follows K2 of this one directly, and EVERY output of this call except the error images is complete only then.
    // Only with device-resident arguments: a host destination is copied back at the end of this call, and a host `perm` / `gt` lives in a staging
    // slot that the next call reuses.
    const int mode = pi_mode(c, perm, gt_jp6_or_null);
    const bool defer = mode != 0;
    const int b = (int)(c->pi_calls++ & 1u);
    c->pi_open = false;  // a begin without its finish is abandoned by a whole call
    if (!d_scores) {
        if (mode == 2) {
            HIP_TRY(c, c->pi_scores[b].reserve((size_t)N * sizeof(double)));
            d_scores = c->pi_scores[b].as<double>();
        } else {
            DevBuf& s = next_slot(c);
            HIP_TRY(c, s.reserve((size_t)N * sizeof(double)));
            d_scores = s.as<double>();
        }
    }
    const int tiles = dk::reproject_num_pixel_tiles(c->F.P);
    DevBuf& part = mode == 2 ? c->pi_soft[b] : c->soft_part;
    HIP_TRY(c, c->staged.reserve((size_t)N * dk::POSE_STRIDE * sizeof(float)));
    HIP_TRY(c, part.reserve((size_t)tiles * N * sizeof(float)));
    const int tk = pi_tail_index(mode, b, N, (long long)P);
    ARG_TRY(pi_tail_setup(c, mode, tk));
    // the call before the previous one used this half of the alternating buffers (and, by the caller's contract, possibly these output arrays): its K3
    // on the tail stream must have read them.  Two K2 launches have run since -- the event has long completed
    if (mode == 2 && c->pi_scored_rec[b]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pi_scored[b], 0));
    // processImage (core/cnn_softam.h:960-1179) for every frame of the batch, one launch per stage:
    //   K1 sample + P3P (:1010-1060)  ->  K2 error images + soft-inlier sums (:1067-1072)  ->  scores  ->  K3 softmax / entropy / soft-argmax
    //   (:1078-1094)  ->  K6 the refinement loop, one wave per frame (:1099-1154)  ->  K7 maxLoss against each frame's ground truth (:1160-1179)
    HIP_TRY(c, dk::sample(c->stream, N, seed, nullptr, c->F, (int)thr, max_tries, d_poses, d_sets, d_ok, c->staged.as<float>(), Nf, c->k1));
    int used = 0;
    if (c->k2_wait) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->k2_wait, 0));
    // mode 2: the tail starts behind K2.  The event it waits for rides on K2's own dispatch packet (hipExtLaunchKernelGGL's stop event) -- an event
    // RECORD between K2 and the next call's K1 is a packet of its own and a ~7 us bubble on the stream that bounds the loop (rank step of configs[3]:
    // 18.6 us from the end of K2 to the start of the next K1 with the record and the wait below, profiles/r04_rank_timeline_mode2.txt)
    hipEvent_t k2_done = nullptr;
    {
        HIP_TRY(c, k2_records_lo(c, c->stream, N, d_poses));
        ProfScope ps(c, 0, true);
        dk::K2Opts o = ps.k2(d_poses);
        if (mode == 2) {
            if (!o.ev_stop) o.ev_stop = c->pi_k2done;
            k2_done = o.ev_stop;
        }
        HIP_TRY(c, dk::reproject(c->stream, N, c->staged.as<float>(), c->F, clampv, d_err, tau, beta, part.as<float>(), o, &used, Nf));
        ps.commit();
    }
    if (c->k2_record) HIP_TRY(c, hipEventRecord(c->k2_record, c->stream));
    hipStream_t ts = c->stream;
    if (mode == 2) {
        // this tail does not depend on the previous call's (different arrays: the mode's contract); it follows the tail of the call two back, whose
        // arrays it may have been given again -- in stream order, or through that stream's completion event when the two used different streams
        ts = c->tail[tk];
        if (c->pi_tail_of[b] >= 0 && c->pi_tail_of[b] != tk) HIP_TRY(c, hipStreamWaitEvent(ts, c->tail_done[c->pi_tail_of[b]], 0));
        c->pi_tail_of[b] = tk;
        HIP_TRY(c, hipStreamWaitEvent(ts, k2_done, 0));
        HIP_TRY(c, score_tail(ts, hyps_per_frame, frames, used, part.as<float>(), d_scores, scale, d_w, d_ent, d_poses, d_avg));
        HIP_TRY(c, hipEventRecord(c->pi_scored[b], ts));
        c->pi_scored_rec[b] = true;
    } else {
        HIP_TRY(c, dk::reduce_soft(c->stream, N, used, part.as<float>(), d_scores));
        // the previous call's tail reads the soft-argmax poses that K3 is about to overwrite.  After a batch it finished long ago (K1 and K2 ran since);
        // in a loop of single images it is the longer chain, and the reduction above runs while the stream would otherwise wait for it
        join_tail(c);
        HIP_TRY(c, dk::softmax(c->stream, hyps_per_frame, d_scores, scale, d_w, d_ent, d_poses, d_avg, frames));
        if (defer) {
            HIP_TRY(c, hipEventRecord(c->tail_go, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->tail[0], c->tail_go, 0));
            ts = c->tail[0];
        }
    }
    ARG_TRY(pi_refine_tail(c, ts, defer, tk, frames, d_avg, d_perm, steps, max_inl, min_inl, thr, d_ref, d_maps, d_sd, d_gt, d_out4));
    return end_call(c);
}

// ---- the score-CNN seam of the batched fast path: dsac_process_images cut between K2 and K3 -----------------------------------------------------
// core/cnn_softam.h:1066-1078 is  getDiffMap x N -> forward(diffMaps) -> softMax : begin leaves the error images of every frame in HBM, the caller's
// score model (the reference's score CNN; any device code on the context's stream) turns them into frames x hyps_per_frame scores, finish continues
// with K3 -> K6 -> K7.  The deferral modes of dsac_process_images apply to the pair (the tails start in finish).
int dsac_process_images_begin(dsac_ctx* c, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clampv, float tau, float beta, double* poses,
                              int32_t* sets_out, uint8_t* ok, float* err, double* soft_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_process_images_begin: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_process_images_begin: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (hyps_per_frame <= 0 || (frames > 1 && hyps_per_frame % dk::K2_NF_MULTIPLE != 0))
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images_begin: hyps_per_frame must be positive (a multiple of %d for a frame batch), got %d", dk::K2_NF_MULTIPLE,
                    hyps_per_frame);
    if ((long long)hyps_per_frame * frames > (1ll << 24)) return fail(c, DSAC_ERR_INVALID, "dsac_process_images_begin: too many hypotheses");
    if (!poses || !sets_out || !ok || (!err && !soft_or_null))
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images_begin: poses / sets_out / ok and at least one of err / soft must be non-NULL");
    if (max_tries <= 0 || c->F.P < 4) return fail(c, DSAC_ERR_INVALID, "dsac_process_images_begin: max_tries > 0 and a frame of at least 4 cells needed");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c, /*keep_tail=*/c->pi_defer_tail != 0);
    const size_t P = (size_t)c->F.P;
    const int N = hyps_per_frame * frames, Nf = frames > 1 ? hyps_per_frame : 0;
    double *d_poses, *d_soft;
    int32_t* d_sets;
    uint8_t* d_ok;
    float* d_err;
    ARG_TRY(out_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, sets_out, (size_t)N * 4, &d_sets));
    ARG_TRY(out_arg(c, ok, (size_t)N, &d_ok));
    ARG_TRY(out_arg(c, err, (size_t)N * P, &d_err));
    ARG_TRY(out_arg(c, soft_or_null, (size_t)N, &d_soft));
    const int b = (int)(c->pi_calls++ & 1u);
    c->pi_open = true; c->pi_open_b = b; c->pi_open_N = hyps_per_frame; c->pi_open_frames = frames;
    // "pi_defer_tail" 2: K3 of the call two back (same half of the caller's alternating arrays) read the poses K1 is about to overwrite
    if (c->pi_defer_tail == 2 && c->pi_scored_rec[b]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pi_scored[b], 0));
    const int tiles = dk::reproject_num_pixel_tiles(c->F.P);
    HIP_TRY(c, c->staged.reserve((size_t)N * dk::POSE_STRIDE * sizeof(float)));
    // the per-tile sums are reduced in stream order below, so the context's one buffer serves every call.  Error images only on a big launch still take
    // the kernel form with the sigmoid arithmetic (dsac_reproject: it is the faster store schedule) and drop the sums
    const bool fused_for_err = !d_soft && c->k2.variant < 0 && !(c->k2.flags & (1 << 24)) && (double)N * (double)P * 4.0 > 1.0e9;
    float* d_part = nullptr;
    if (d_soft || fused_for_err) {
        HIP_TRY(c, c->soft_part.reserve((size_t)tiles * N * sizeof(float)));
        d_part = c->soft_part.as<float>();
    }
    if (fused_for_err && !(beta > 0.f)) { tau = 10.f; beta = 0.5f; }
    HIP_TRY(c, dk::sample(c->stream, N, seed, nullptr, c->F, (int)thr, max_tries, d_poses, d_sets, d_ok, c->staged.as<float>(), Nf, c->k1));
    int used = 0;
    if (c->k2_wait) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->k2_wait, 0));
    {
        HIP_TRY(c, k2_records_lo(c, c->stream, N, d_poses));
        ProfScope ps(c, 0, true);
        HIP_TRY(c, dk::reproject(c->stream, N, c->staged.as<float>(), c->F, clampv, d_err, tau, beta, d_part, ps.k2(d_poses), &used, Nf));
        ps.commit();
    }
    if (c->k2_record) HIP_TRY(c, hipEventRecord(c->k2_record, c->stream));
    if (d_soft) HIP_TRY(c, dk::reduce_soft(c->stream, N, used, d_part, d_soft));
    return end_call(c);
}

int dsac_process_images_finish(dsac_ctx* c, int hyps_per_frame, const double* scores, double scale, const int32_t* perm, int steps, int max_inl, int min_inl,
                               float thr, const double* gt_jp6_or_null, const double* poses, double* w, double* entropy, double* avg6, double* ref6,
                               int32_t* steps_done, int32_t* inlier_maps_or_null, double* out4_or_null) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_process_images_finish: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_process_images_finish: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;
    if (!c->pi_open || c->pi_open_N != hyps_per_frame || c->pi_open_frames != frames)
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images_finish: no dsac_process_images_begin of %d frame(s) x %d hypotheses is open on this context", frames,
                    hyps_per_frame);
    if (!scores || !perm || !poses || !w || !entropy || !avg6 || !ref6 || !steps_done || steps < 0)
        return fail(c, DSAC_ERR_INVALID, "dsac_process_images_finish: NULL argument or negative step count");
    if ((out4_or_null != nullptr) != (gt_jp6_or_null != nullptr)) return fail(c, DSAC_ERR_INVALID, "dsac_process_images_finish: out4 and gt_jp6 go together");
    if (max_inl < 1 || max_inl > 256 || min_inl < 0) return fail(c, DSAC_ERR_INVALID, "dsac_process_images_finish: need 1 <= max_inl <= 256 (got %d), min_inl >= 0", max_inl);
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c, /*keep_tail=*/c->pi_defer_tail != 0);
    const size_t P = (size_t)c->F.P;
    const int N = hyps_per_frame * frames;
    const int32_t* d_perm;
    const double *d_gt, *d_scores, *d_poses;
    double *d_w, *d_ent, *d_avg, *d_ref, *d_out4;
    int32_t *d_sd, *d_maps;
    ARG_TRY(in_arg(c, perm, (size_t)steps * P, &d_perm));
    ARG_TRY(in_arg(c, gt_jp6_or_null, (size_t)frames * 6, &d_gt));
    ARG_TRY(in_arg(c, scores, (size_t)N, &d_scores));
    ARG_TRY(in_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(out_arg(c, w, (size_t)N, &d_w));
    ARG_TRY(out_arg(c, entropy, (size_t)frames, &d_ent));
    ARG_TRY(out_arg(c, avg6, (size_t)frames * 6, &d_avg));
    ARG_TRY(out_arg(c, ref6, (size_t)frames * 6, &d_ref));
    ARG_TRY(out_arg(c, steps_done, (size_t)frames, &d_sd));
    ARG_TRY(out_arg(c, inlier_maps_or_null, (size_t)frames * P, &d_maps));
    ARG_TRY(out_arg(c, out4_or_null, (size_t)frames * 4, &d_out4));
    // the deferral needs every argument in HBM (pi_mode); scores and poses too -- a staged copy lives in a slot the next call reuses
    const int mode = (is_device_ptr(scores, c) && is_device_ptr(poses, c)) ? pi_mode(c, perm, gt_jp6_or_null) : 0;
    const bool defer = mode != 0;
    const int b = c->pi_open_b;
    c->pi_open = false;
    const int tk = pi_tail_index(mode, b, N, (long long)P);
    ARG_TRY(pi_tail_setup(c, mode, tk));
    hipStream_t ts = c->stream;
    if (mode == 2) {
        // K3 as well goes to the tail stream: it starts when the scores are there (an event on the context's stream: the score model ran on it), the
        // next begin's K1 follows the score model without waiting for K3.  Ordering against earlier tails as in dsac_process_images
        ts = c->tail[tk];
        if (c->pi_tail_of[b] >= 0 && c->pi_tail_of[b] != tk) HIP_TRY(c, hipStreamWaitEvent(ts, c->tail_done[c->pi_tail_of[b]], 0));
        c->pi_tail_of[b] = tk;
        HIP_TRY(c, hipEventRecord(c->tail_go, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(ts, c->tail_go, 0));
        HIP_TRY(c, dk::softmax(ts, hyps_per_frame, d_scores, scale, d_w, d_ent, d_poses, d_avg, frames));
        HIP_TRY(c, hipEventRecord(c->pi_scored[b], ts));
        c->pi_scored_rec[b] = true;
    } else {
        join_tail(c);  // the previous call's tail reads the soft-argmax poses that K3 is about to overwrite
        HIP_TRY(c, dk::softmax(c->stream, hyps_per_frame, d_scores, scale, d_w, d_ent, d_poses, d_avg, frames));
        if (defer) {
            HIP_TRY(c, hipEventRecord(c->tail_go, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->tail[0], c->tail_go, 0));
            ts = c->tail[0];
        }
    }
    ARG_TRY(pi_refine_tail(c, ts, defer, tk, frames, d_avg, d_perm, steps, max_inl, min_inl, thr, d_ref, d_maps, d_sd, d_gt, d_out4));
    return end_call(c);
}

int dsac_join_tail(dsac_ctx* c) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_join_tail: ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    join_tail(c);
    return DSAC_OK;
}

int dsac_tail_wait(dsac_ctx* c, void* hip_stream) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_tail_wait: ctx is NULL");
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    if (s == c->stream) return fail(c, DSAC_ERR_INVALID, "dsac_tail_wait: that is the context's own stream (use dsac_join_tail)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->tail_pending[0] || c->tail_pending[1]) {
        // a tail started behind K2 / K3 of its dsac_process_images call, which was behind everything the context's stream held then: the completion
        // events of the tails in flight cover their calls and all earlier work -- no marker has to be put into the context's stream (a record between
        // two steps costs a bubble)
        for (int k = 0; k < 2; k++)
            if (c->tail_pending[k]) HIP_TRY(c, hipStreamWaitEvent(s, c->tail_done[k], 0));
        return DSAC_OK;
    }
    if (!c->xs_event) HIP_TRY(c, hipEventCreateWithFlags(&c->xs_event, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->xs_event, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(s, c->xs_event, 0));
    return DSAC_OK;
}

int dsac_path1_and_softmax_backward(dsac_ctx* c, int N, const double* v6, const double* w, const double* poses, const int32_t* sets,
                                    const double* dpnp, double* grad_xyz, double* g) {
    if (!c) return fail(nullptr, DSAC_ERR_INVALID, "dsac_path1_and_softmax_backward: ctx is NULL");
    if (!c->have_frame) return fail(c, DSAC_ERR_NO_FRAME, "dsac_path1_and_softmax_backward: no frame set");
    const int frames = c->F.frames > 1 ? c->F.frames : 1;  // frame batch: N = frames x hypotheses per frame, v6 frames x 6, grad_xyz frames x H*W x 3
    if (N <= 0 || N % frames != 0 || !v6 || !w || !poses || !g) return fail(c, DSAC_ERR_INVALID, "dsac_path1_and_softmax_backward: NULL argument or N not frames x hypotheses per frame");
    if ((grad_xyz != nullptr) != (dpnp != nullptr) || (grad_xyz && !sets))
        return fail(c, DSAC_ERR_INVALID, "dsac_path1_and_softmax_backward: grad_xyz, dpnp and sets go together");
    HIP_TRY(c, hipSetDevice(c->device));
    begin_call(c);
    const size_t P = (size_t)c->F.P;
    const double *d_v6, *d_w, *d_poses, *d_dpnp;
    const int32_t* d_sets;
    double *d_grad, *d_g;
    ARG_TRY(in_arg(c, v6, (size_t)frames * 6, &d_v6));
    ARG_TRY(in_arg(c, w, (size_t)N, &d_w));
    ARG_TRY(in_arg(c, poses, (size_t)N * 6, &d_poses));
    ARG_TRY(in_arg(c, sets, (size_t)N * 4, &d_sets));
    ARG_TRY(in_arg(c, dpnp, (size_t)N * 72, &d_dpnp));
    ARG_TRY(out_arg(c, grad_xyz, (size_t)frames * P * 3, &d_grad, /*preload=*/true));
    ARG_TRY(out_arg(c, g, (size_t)N, &d_g));
    HIP_TRY(c, dk::path1_softmax_backward(c->stream, N / frames, c->F.P, d_v6, d_w, d_poses, d_sets, d_dpnp, d_grad, d_g, 1.0, frames));
    return end_call(c);
}

}  // extern "C"

// kernels.h -- host-callable launchers of the gfx950 kernels (one .hip translation unit per group).
// All pointers are DEVICE pointers; every launcher only enqueues work on `st` and returns the HIP status
// of the last launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dk {

struct FrameDev {
    const float* xyz;  // P x 3 (mm)
    const float* uv;   // P x 2 or nullptr (implicit grid u = x, v = y)
    int H, W, P;
    float fx, fy, cx, cy;
    // frame batch (dsac_set_frames): `frames` maps of the same geometry back to back, frame f at xyz + f * xyz_stride
    // (floats); uv is shared (uv_stride == 0) or per frame.  Only K1 (random sampling), K2 and K3 look at frames > 0.
    int frames = 1;
    long long xyz_stride = 0, uv_stride = 0;
    int seed_stride = 1;  // frame f of a batch draws from the random stream of seed + f * seed_stride (dsac_set_option "seed_stride")
};

// Staged pose record used by K2, 12 floats (48 B, three float4 rows) per hypothesis:
//   [0..3]  fx*R0 | fx*tx      [4..7] fy*R1 | fy*ty      [8..11] R2 | tz
constexpr int POSE_STRIDE = 12;

// ---- k_forward.hip ---------------------------------------------------------------------------------
// fp64 Rodrigues of N cv poses -> staged float records.
hipError_t pose_prep(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged);
hipError_t pose_prep_lo(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged_lo);  // what the float records leave behind
// the records of the exact-transform form (k2_flags bit 28): pose_split_bytes(N) bytes; available for focal lengths up to 2^10 (pose_split_exponent <= 10)
size_t pose_split_bytes(int N);
int pose_split_exponent(const FrameDev& F);
bool pose_split_available(const FrameDev& F);
hipError_t pose_prep_split(hipStream_t st, int N, const double* poses, const FrameDev& F, void* split);

// K2.  err (N x P) and/or soft partials.  soft_part must hold reproject_num_pixel_tiles(P) * N floats.
// Launch knobs of K2; they live in the context (read once from the environment in dsac_create), never in process-wide state.
struct K2Opts {
    bool pixel_minor = true;  // block order: pixel tiles innermost (DSAC_K2_ORDER)
    int flags = 0;            // bit0: plain (cached) stores instead of non-temporal (DSAC_K2_FLAGS)
    int variant = -1;         // -1 = auto policy, otherwise a fixed kernel form (DSAC_K2_VARIANT), see reproject()
    bool exact_auto = true;   // "k2_exact_auto" (DSAC_K2_EXACT_AUTO): the auto policy (variant -1, no other arithmetic flag) takes the exact-transform form wherever it
                              // applies -- round 6: the default K2 is the form that holds every stated tolerance; 0 = the fp32 matrix-core forms
    int diag = 0;             // "k2_diag": diagnostic switches of the precise form (which fp32 step costs what; k_reproject_prec), 0 = none
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // per call: timing events attached to the K2 dispatch itself (profiling), else null
    const float* staged_lo = nullptr;  // per call: the low parts of the staged records (pose_prep_lo), for flags bit 27
    const void* split = nullptr;      // per call: the split fp16 records (pose_prep_split), for flags bit 28
    const double* poses64 = nullptr;  // per call: the cv poses (N x 6 doubles) the staged records were made from -- the precise form (flags bit 25) works from these
};
constexpr int K2_FLAG_RECLO = 1 << 27;    // k2_flags: pose records in two pieces -- the low parts through fp16 matrix-core instructions chained onto the fp32 ones
constexpr int K2_FLAG_STORE_ONLY = 1 << 1;  // k2_flags: store schedule only (measurement)
constexpr int K2_FLAG_EXACT = 1 << 28;    // k2_flags: exact transform -- split fp16 records through the fp16 matrix core, the camera-frame point rounded to float once (round 6)
constexpr int K2_FLAG_PRECISE = 1 << 25;  // k2_flags: the fp64 projection of the reference (k_reproject_prec) instead of the fp32 matrix-core transform
// does a K2 launch with these options want the split records?  (bit 28 asks for them; the auto policy wants them unless another arithmetic form or a measurement flag is set)
inline bool k2_wants_exact(const K2Opts& o) {
    if (o.flags & K2_FLAG_EXACT) return true;
    return o.exact_auto && o.variant < 0 && !(o.flags & (K2_FLAG_PRECISE | K2_FLAG_RECLO | K2_FLAG_STORE_ONLY | (1 << 26)));
}
int reproject_num_pixel_tiles(int P);  // upper bound over both code paths
// *tiles_used receives the number of pixel tiles actually written to soft_part (<= reproject_num_pixel_tiles).
// Nf: hypotheses per frame of a frame batch (hypothesis h scores frame h / Nf; Nf must be a multiple of K2_NF_MULTIPLE = 128 then: no
// hypothesis tile of any kernel form may straddle two frames); 0 = one frame.
constexpr int K2_NF_MULTIPLE = 128;
bool reproject_variant_known(int variant);  // -1 (auto) or a value reproject() has a kernel form for
hipError_t reproject(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float tau, float beta,
                     float* soft_part, const K2Opts& opts, int* tiles_used, int Nf = 0);
// soft[h] = sum over pixel tiles of soft_part[tile][h]   (double, deterministic)
hipError_t reduce_soft(hipStream_t st, int N, int tiles, const float* soft_part, double* soft);
// K3.
// frames > 1: one independent softmax per consecutive group of N scores (outputs entropy[frames], avg6[frames][6])
hipError_t softmax(hipStream_t st, int N, const double* scores, double scale, double* w, double* entropy, const double* poses, double* avg6,
                   int frames = 1);

// ---- k_sample.hip ----------------------------------------------------------------------------------
// Nf > 0 (frame batch): hypothesis h belongs to frame h / Nf and draws from the stream of (seed + frame, h % Nf), i.e. exactly
// what a single-frame call with seed + frame would draw.
// Launch knobs of K1 (context state like K2Opts): waves per workgroup (DSAC_K1_WPB in {1, 4, 8}), wave priority (DSAC_K1_PRIO 0..3),
// hypotheses per wave (DSAC_K1_HPW in {1, 2, 4}), OpenCV's Jacobi sweeps for the alignment of the P3P triangle instead of the closed form (DSAC_K1_HORN).
struct K1Opts {
    int wpb = 1, prio = 3, hpw = 1, minw = 1;  // minw: minimum waves per SIMD of the register allocation (DSAC_K1_MINW)
    bool share_always = false;  // share beyond 1024 hypotheses as well (DSAC_K1_SHARE < 0)
    int share = 4;  // waves of a workgroup that help each other's unfinished hypotheses (k_sample_shared; 0 / 1 = off, 4, 8; DSAC_K1_SHARE)
    bool horn = false;
    int wide = -1;  // waves per hypothesis of the one-lane-per-attempt form for few hypotheses: -1 auto (2 up to 512 hypotheses), 0 off, 2, 4 (4: up to 256) (DSAC_K1_WIDE)
    int rl = 1;  // lanes per attempt: 4 = one per quartic root, 1 = one lane per attempt with its roots in sequence (DSAC_K1_RL)
};
hipError_t sample(hipStream_t st, int N, uint64_t seed, const int32_t* sets_in, const FrameDev& F, int thr_int, int max_tries,
                  double* poses, int32_t* sets_out, uint8_t* ok, float* staged_or_null = nullptr, int Nf = 0,
                  const K1Opts& opts = K1Opts());  // staged: K2 records (N x 12)
// K1 in the reference's own random stream (round 6; core/thread_rand.cpp:40-69, core/cnn_softam.h:1010-1060; refstream.h): T generators std::mt19937(seed + t)
// on the device; a window = A attempts per stream parsed, evaluated and handed to the stream's next hypotheses in order.
struct RefStreamState { uint32_t mt[624]; uint32_t idx; uint32_t pad[3]; };
inline int refstream_window_outputs(int A) { return 9 * A + 128; }  // raw outputs generated per window (an attempt takes 8, rarely more)
size_t refstream_window_bytes(int T, int A);
hipError_t refstream_init(hipStream_t st, RefStreamState* states, uint32_t seed, int T);
hipError_t refstream_discard(hipStream_t st, RefStreamState* states, int t, unsigned long long n);
// first / served / need / parsed [T] int32, consumed [T] uint64, attempts [T] int64: device arrays (served / need / consumed / attempts are updated)
hipError_t refstream_window(hipStream_t st, RefStreamState* states, int T, int A, int mode, void* scratch, const FrameDev& F, int thr_int, const int32_t* first,
                            int32_t* served, int32_t* need, int32_t* parsed, unsigned long long* consumed, long long* attempts, double* poses, int32_t* sets_out,
                            uint8_t* ok, float* staged);
hipError_t refstream_unserved(hipStream_t st, int T, const int32_t* first, const int32_t* served, const int32_t* need, const FrameDev& F, double* poses, int32_t* sets_out,
                              uint8_t* ok, float* staged);
// Nf (frame batch): hypotheses per frame, hypothesis h reads frame h / Nf
hipError_t dpnp(hipStream_t st, int N, const int32_t* sets, const FrameDev& F, float eps, double* J, int Nf = 0);

// ---- k_backward.hip --------------------------------------------------------------------------------
// jp-convention staged records for the backward pass, BWD_STRIDE floats per hypothesis (see k_backward.hip).
constexpr int BWD_STRIDE = 24;  // six float4 per hypothesis, already sign-folded and pair-packed for K4's packed-fp32 chains
// Launch plan of the K4 main pass: which kernel form, its hypothesis tile and the sizes of the two partial-sum buffers.
//   variant 0 = VALU form (lane = 8 pixels, wave reduction of the 12 sums per hypothesis), 1 .. 5 = matrix-core form with 2 / 4 / 5 / 6 / 3 16-pixel
//   chunks per wave (lane = one hypothesis x 4 consecutive pixels), -1 = auto.  Maps that cannot be read as 16-byte vectors use variant 0.
struct K4Plan {
    int variant;  // resolved form
    int HT, NT;   // hypotheses per workgroup, number of hypothesis tiles (rows of grad_part)
    int rows;     // rows of G12_part the launch writes
    int glayers;  // layers of grad_part per hypothesis tile (matrix-core form: 2 -- a pixel tile's hypothesis groups may be split between two workgroups)
    bool fused = false;   // round 4: no backward_prep launch -- the main pass derives its records from the poses, the finish kernel dR/drod; G12_part [hyp][row][12]
    bool direct = false;  // ... and (one hypothesis tile) the main pass adds its gradient straight into grad_xyz: no grad_part, no gradient reduction launch
    int Nf = 0;           // frame batch: hypotheses per frame (= HT), 0 = one frame
};
// Nf > 0 (frame batch): N = frames x Nf hypotheses, one hypothesis tile per frame (Nf a multiple of 16, <= 256), gradient per frame (grad_xyz frames x P x 3);
// plan.variant == 0 then means: not available for a batch
K4Plan backward_plan(int N, const FrameDev& F, const float* d_err, int variant, int Nf = 0);
bool backward_variant_known(int variant);  // -1 (auto), 0 .. 5, or a form + 10 * tile code + 100 * workgroups per CU (see backward_plan)
constexpr int BWD_DRDH = 54;  // per hypothesis: dR/drod (27) and Omega_i = (dR/drod_i) R^T (27)
hipError_t backward_prep(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged_bwd, double* dRdH /*N x BWD_DRDH*/);
// K4 main pass.  d_err (N x P) or nullptr with g (N doubles) for the soft-inlier score.
//   grad_part : [hyp_tiles][P*3] floats       G12_part : [partial rows][N][12] floats
// poses (cv, N x 6) / grad_xyz (P x 3 fp64, accumulated into) / flags: used when plan.fused / plan.direct (staged_bwd and grad_part may then be nullptr)
hipError_t score_backward(hipStream_t st, int N, const float* staged_bwd, const FrameDev& F, const float* d_err, const double* g,
                          float clampv, float tau, float beta, float* grad_part, float* G12_part, const K4Plan& plan, const double* poses = nullptr,
                          double* grad_xyz = nullptr, unsigned flags = 0);
// Epilogue: grad_xyz (P x 3 double) += sum over hyp tiles; then per hypothesis G6 = [G9 * dRdH, G3],
// S = G6 * dPNP_h, scatter-add S to the 4 support pixels.
hipError_t score_backward_finish(hipStream_t st, int N, const FrameDev& F, const float* grad_part, int hyp_tiles, const float* G12_part,
                                 int pixel_tiles, const double* dRdH, const double* dpnp, const int32_t* sets, unsigned flags,
                                 double* grad_xyz, double* G6_scratch, const float* rec_if_e_based = nullptr, const double* poses_if_fused = nullptr, int Nf = 0);
// hyp_tiles == 0: no gradient reduction (plan.direct); poses_if_fused: the cv poses when plan.fused (dRdH and rec_if_e_based are then unused)
// rec_if_e_based: the BWD records when G12_part holds the matrix-core form's E-based sums (K4Plan.variant > 0), else nullptr
// K4 parity mode (fp64, the reference's evaluation order; flags bit 0 = transposed columns, bit 2 = rotation write-back); jac_scratch: N x P*3 doubles
hipError_t score_backward_parity(hipStream_t st, int N, const double* poses, const FrameDev& F, const float* d_err, const double* dpnp, const int32_t* sets,
                                 unsigned flags, double* jac_scratch, double* grad_xyz, double* G6_out);
// frames > 1: N hypotheses PER FRAME, v6 frames x 6, grad_xyz frames x P x 3, one workgroup per frame
hipError_t path1_softmax_backward(hipStream_t st, int N, int P, const double* v6, const double* w, const double* poses, const int32_t* sets,
                                  const double* dpnp, double* grad_xyz, double* g, double g_scale = 1.0, int frames = 1);
// grad[obj_pixels[i]] += dL . J_obj[i] for i < min(*n_obj, cap), v6 = dL . J_hyp
// frames > 1: every array holds one slice per frame (dL 6, J_hyp 36, obj_pixels px_stride, J_obj cap*18, n_obj 1, grad_xyz P*3, v6 6)
hipError_t path1_assemble(hipStream_t st, const double* dL, const double* J_hyp, const int32_t* obj_pixels, const double* J_obj, const int32_t* n_obj,
                          int cap, int P, double* grad_xyz, double* v6, int frames = 1, int px_stride = 0);

// ---- k_refine.hip ----------------------------------------------------------------------------------
// inlier_map: map_stride == 0 -> H*W counters of problem 0 only; map_stride == H*W -> one map per problem
// per_frame > 0 (frame batch): problem b refines against frame b / per_frame (F.xyz_stride / F.uv_stride)
hipError_t refine(hipStream_t st, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                  const int32_t* pert_px_c, const float* pert_value, const FrameDev& F, double* out_poses, int32_t* inlier_map,
                  int32_t* steps_done, int map_stride = 0, int per_frame = 0, const double* loss_gt_jp6 = nullptr, double* loss_out4 = nullptr,
                  int waves_per_problem = 0);
// many problems with long walks (the DSAC variant on big maps): a refinement step as two launches, a scan of the step's pre-permuted cells + LM (k_refine.hip); the
// same results bit for bit.  refine_split_applies: >= 32 problems, >= 16 384 cells, no perturbation, no fused loss, "k6_waves" 0
size_t refine_split_scratch_bytes(int B, int steps, int frames, int P, int max_inl);
void refine_scan_tune(int v);  // experiments ("k6_scan_tune")
bool refine_split_applies(int B, const FrameDev& F, const int32_t* pert_px_c, const double* loss_out4, int waves_per_problem);
hipError_t refine_split(hipStream_t st, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr, const FrameDev& F,
                        double* out_poses, int32_t* inlier_map, int32_t* steps_done, int map_stride, int per_frame, void* scratch, int exact_only = 0);
// exact_only ("k6_walk_exact"): the walk's fp32 filter off -- every cell by the fp64 residual (A/B; the decisions are the same)
// waves_per_problem: 0 = by the problem count (1 for a single problem, 4 up to 512 problems, 2 up to 1 024, else 1), 1 / 2 / 4 / 8 = fixed ("k6_waves"); the same results bit for bit
// loss_gt_jp6 (B x 6) / loss_out4 (B x 4): maxLoss of every refined pose against its ground truth in the same launch (K7's arithmetic, loss_math.h)
// inlier_maps[h][set cell] = 0 for the 4 cells of every hypothesis' minimal set (core/cnn.h:1208-1214)
hipError_t zero_set_cells(hipStream_t st, int N, const int32_t* sets, int P, int32_t* inlier_maps);
// DSAC-variant replica plan (core/cnn.h:854-990 dRefine): 18 replicas perturb the first three points of the minimal set,
// 6 per selected inlier cell follow; every replica's start pose is P3P of the set read from the perturbed map.
// scratch: M * refine_fd_plan_scratch_ints(F) int32 (nullptr / small maps: one workgroup per hypothesis scans its map)
hipError_t refine_fd_plan_set(hipStream_t st, const int32_t* set4, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_obj, int cap,
                              double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels, int32_t* n_obj, int M = 1,
                              int32_t* scratch = nullptr, const int32_t* frame_of = nullptr);
// frame_of (M int32, device) with a frame batch: hypothesis m lives in frame frame_of[m] (its minimal set, inlier map and replicas read that frame's map)
// M hypotheses at once (sets M x 4, inlier maps M x H*W, 18 + 6*cap replicas each): one launch of M * (18 + 6*cap) waves
hipError_t refine_fd_run_set(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                             int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out, int M = 1,
                             const int32_t* frame_of = nullptr);
hipError_t refine_fd_finish_set(hipStream_t st, const double* rep_out, const int32_t* n_obj, int cap, int skip, float eps_obj, double* J_set /*M x 6 x 9*/,
                                double* J_obj, int M = 1);
// builds the replica list of dRefineHyp/dRefineObj on device, see k_refine.hip
// scratch: refine_fd_plan_scratch_ints(F) int32 of device memory (0: not needed, small map) -- large maps are planned by two tiled launches
size_t refine_fd_plan_scratch_ints(const FrameDev& F);
// frames > 1 (frame batch): one replica list of 12 + 6*cap entries per frame (init_pose frames x 6, inlier_map frames x H*W, obj_pixels frames x px_stride,
// n_obj frames, scratch frames x refine_fd_plan_scratch_ints); perturbed values are read from the frame's own map
hipError_t refine_fd_plan(hipStream_t st, const double* init_pose, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_hyp,
                          float eps_obj, int cap, double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels,
                          int32_t* n_obj, int32_t* scratch = nullptr, int frames = 1, int px_stride = 0);
// launches 12 + 6*cap replica waves; those beyond 12 + 6*n_obj[0] exit immediately
hipError_t refine_fd_run(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                         int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out, int frames = 1);
hipError_t refine_fd_finish(hipStream_t st, const double* rep_out, const int32_t* n_obj, int cap, int skip, float eps_hyp, float eps_obj,
                            double* J_hyp, double* J_obj, int frames = 1);

// ---- k_loss.hip ------------------------------------------------------------------------------------
// gt_stride: 0 = one ground truth (6 doubles) for all estimates, 6 = one per estimate
// gt_group: estimates b share the ground truth of index b / gt_group (a frame batch of the DSAC variant: N estimates per frame)
hipError_t pose_loss(hipStream_t st, int B, const double* est_cv6 /*B x 6*/, const double* gt_jp6, double* out4 /*B x 4*/, double* J6 /*B x 6*/,
                     int gt_stride = 0, int gt_group = 1);

// DSAC variant: draw (core/cnn.h:102-127; u < 0: argmax), expectedMaxLoss (:137-150), the score gradients of dSMScore (:737-742); losses[i * loss_stride]
hipError_t dsac_select(hipStream_t st, int N, const double* w, const double* losses, int loss_stride, double u, double eps, int32_t* hyp_idx, double* expected,
                       double* g);
// the same for `frames` images in one launch (a workgroup per image): w / losses / g [frames][N], u / hyp_idx / expected [frames]; u on the device
hipError_t dsac_select_frames(hipStream_t st, int frames, int N, const double* w, const double* losses, int loss_stride, const double* u, double eps,
                              int32_t* hyp_idx, double* expected, double* g);

// ---- k_patches.hip ---------------------------------------------------------------------------------
// n patches of 3 x patch x patch floats (patch, channel, row, column) around sampling_xy[i] = (x, y) of an H x W x 3 BGR image
hipError_t gather_patches(hipStream_t st, const uint8_t* bgr, int H, int W, const int32_t* sampling_xy, int n, int patch, float* out, int32_t* skipped);

}  // namespace dk

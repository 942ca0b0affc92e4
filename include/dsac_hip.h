/* dsac_hip.h -- C ABI of libdsac_hip.so, the MI355X (gfx950) DSAC hypothesis-scoring engine.
 *
 * This is the drop-in boundary for the soft-argmax hot path of cvlab-dresden/DSAC.  The reference has no
 * FFI seam for its geometry: every function below is a free function defined in a header and compiled
 * into each executable (core/cnn_softam.h is #included by core/train_ransac_softam.cpp:39 and
 * core/test_ransac_softam.cpp:37).  The boundary is therefore cut at those function signatures; each
 * export cites the reference function (file:line under /root/reference/core) it replaces.  The C++ host
 * shim dsac_amd/host/ keeps the reference's own names (Hypothesis, getDiffMap, softMax, dPNP, refine,
 * dScore, dLossMax, processImage) on top of this ABI; INTEGRATION.md shows the reference-side binding.
 *
 * Conventions (identical to the reference):
 *   pose      : 6 doubles  rvec[3] (Rodrigues, rad) | tvec[3] (mm), OpenCV convention (jp::cv_trans_t,
 *               core/types.h:91) unless a parameter is explicitly called "jp6" (jp::jp_trans_t as the
 *               6-vector Hypothesis::getRodVecAndTrans returns, core/Hypothesis.cpp:274-289).
 *   frame     : scene-coordinate map H x W x 3 float32 millimetres, row-major (y, x, c) -- the reference's
 *               jp::img_coord_t (int16 mm, core/types.h:43-51) generalised to float; DSAC_FRAME_QUANTISE_INT16
 *               rounds/saturates to the int16 grid on upload (core/cnn_softam.h:265).  Pixel positions
 *               ("sampling", core/cnn_softam.h:283-309) are H x W x 2 float32 (u, v) or NULL for the
 *               implicit full-resolution grid u = x, v = y.  A pixel index is y*W + x.
 *   camera    : fx, fy, cx, cy (core/properties.cpp:308-323: fx = fy = 525, cx = 320, cy = 240).
 *   err image : N x (H*W) float32, hypothesis-major, each image row-major -- the order the reference
 *               pushes diffMaps to the score CNN (core/lua_calls.h:98-104).
 *
 * Memory: every pointer argument may be a host pointer or a device (HIP) pointer; the library detects
 * which (hipPointerGetAttributes).  With device pointers a call only enqueues work on the context's
 * stream and returns; with host pointers it stages through its own scratch and returns after the result
 * has landed.  The caller owns every buffer; the library owns only the context and its device scratch,
 * and retains nothing after return except the frame copied by dsac_set_frame (or borrowed when asked).
 *
 * Errors: every call returns DSAC_OK (0) or a negative dsac_status; nothing throws across the ABI.
 * Per-hypothesis failure (no P3P solution within max_tries) is reported in ok[] with a zero pose, which
 * is the reference's safeSolvePnP behaviour (core/cnn_softam.h:66-71).  NaN policy as the reference
 * (zero Jacobians).
 *
 * Threading: a context is single-threaded; use one context per host thread / stream / GPU.  The library keeps no
 * process-wide mutable state: launch knobs live in the context (dsac_set_option), and the only state outside a context is
 * the calling thread's own last-error string (thread_local; dsac_last_error(NULL) after a failed dsac_create).  Contexts on
 * different host threads may be used concurrently.  There is NO CPU fallback: without a HIP device dsac_create fails with
 * DSAC_ERR_NO_DEVICE.
 *
 * Sampling RNG (shared bit-exactly with the CPU oracle so that minimal sets are identical):
 *   mix64(z): z += 0x9E3779B97F4A7C15; z = (z ^ (z>>30)) * 0xBF58476D1CE4E5B9;
 *             z = (z ^ (z>>27)) * 0x94D049BB133111EB; return z ^ (z>>31);
 *   key(h)          = mix64(seed ^ mix64(h))                       h = hypothesis index
 *   v(h,a,k)        = mix64(key(h) + ((a << 16) | k))                       candidate k of attempt a
 *   cell(h,a,k)     : x = ((v >> 32) * W) >> 32,  y = ((v & 0xffffffff) * H) >> 32   (one 64-bit draw per cell)
 *   attempt a of hypothesis h takes the candidate cells k = 0, 1, 2, ... until 4 distinct ones (core/cnn_softam.h:1021-1039:
 *   duplicates redrawn; more than 32 candidates: the attempt fails); the first accepted attempt in
 *   a = 0,1,2,... wins, which is the reference's while(true) loop made order-independent.
 */
#ifndef DSAC_HIP_H
#define DSAC_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The only symbols libdsac_hip.so exports: the library is built with -fvisibility=hidden, its C++ internals (namespace dk) stay private. */
#if defined(__GNUC__)
#define DSAC_API __attribute__((visibility("default")))
#else
#define DSAC_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsac_ctx dsac_ctx;

typedef enum dsac_status {
    DSAC_OK = 0,
    DSAC_ERR_INVALID = -1,   /* bad argument */
    DSAC_ERR_NO_DEVICE = -2, /* no HIP device / wrong architecture */
    DSAC_ERR_HIP = -3,       /* a HIP runtime call failed; see dsac_last_error */
    DSAC_ERR_NO_FRAME = -4,  /* dsac_set_frame has not been called */
    DSAC_ERR_ALLOC = -5
} dsac_status;

/* dsac_set_frame flags */
#define DSAC_FRAME_QUANTISE_INT16 1u /* xyz <- saturate_cast<short>(round(xyz)), core/cnn_softam.h:265, types.h:43 */
#define DSAC_FRAME_BORROW 2u         /* xyz/uv are device pointers that outlive the frame: use in place, no copy.  Together with
                                        DSAC_FRAME_QUANTISE_INT16 the caller's xyz buffer is rounded IN PLACE (in stream order) */

/* backward flags */
#define DSAC_BWD_PARITY_FP64 2u        /* dsac_score_backward only: the fp64 parity mode -- the reference's own evaluation order in double (one lane per
                                          hypothesis, cells x outer / y inner, per-hypothesis Jacobians summed in index order); N*H*W <= 2^26 */
#define DSAC_BWD_QUIRK_ROT_WRITEBACK 4u /* with DSAC_BWD_PARITY_FP64: quirk 7 -- dProjectdHyp writes the re-derived rotation back into the hypothesis
                                          (core/cnn_softam.h:506-508), which then drifts by round-off from cell to cell; reproduces the reference to 1e-9 */
#define DSAC_BWD_QUIRK_TRANSPOSE 1u /* reproduce core/cnn_softam.h:628,641 (index x*W*3 + y*3); needs H == W */

/* ---- context ---------------------------------------------------------------------------------- */
DSAC_API const char* dsac_version(void);
DSAC_API int dsac_create(dsac_ctx** out, int device);
DSAC_API void dsac_destroy(dsac_ctx* ctx);
DSAC_API const char* dsac_last_error(dsac_ctx* ctx_or_null);
DSAC_API int dsac_set_stream(dsac_ctx* ctx, void* hip_stream); /* adopt an external hipStream_t (e.g. torch's) */
DSAC_API void* dsac_get_stream(dsac_ctx* ctx);
DSAC_API int dsac_synchronize(dsac_ctx* ctx);
/* device facts for reports: CU count, clock (kHz), total memory (bytes), gcnArchName into name[64] */
DSAC_API int dsac_device_info(dsac_ctx* ctx, int* cus, int* clock_khz, uint64_t* mem_bytes, char* name64);
/* Per-context launch knobs (what the reference keeps in its GlobalProperties singleton, core/properties.h, made per-context and explicit).
 * None of them changes a result beyond rounding; -1 / the default is the measured policy.  Unknown keys are DSAC_ERR_INVALID.
 *   "k2_variant"  K2 kernel form: -1 auto; 0-3, 10-13 VALU forms; 20-27 matrix-core forms <hypothesis tile, chunks per wave>
 *   "k2_order"    1 = pixel tiles innermost in K2's block order (default), 0 = hypothesis tiles innermost
 *   "k2_flags"    bit0: cached instead of non-temporal stores; bit1: store schedule only, no arithmetic (ceiling measurement); bits 2-4: cache policy of the error-image stores, 0 = nt (measurement: none of nt/plain/sc1/sc0 sc1/sc1 nt/sc0 differs by more than 1.5 %); bits 8-15: units of 8 KiB unused LDS per workgroup (occupancy cap, experiments); bits 16-19 / 20-23: error-images-only streaming forms idle for that many units of 64 / 16 clocks after a chunk's stores (pacing experiment); bit 24: error images only on a big launch do NOT take the fused kernel; bit 25 (0x2000000): PRECISE -- K2 projects as the reference does, in double (fp64 pose records from the cv poses, fp64 transform and perspective division, one rounding to float of each image-plane difference; core/cnn_softam.h:319-362): residuals within 2e-4 px of the oracle at 640 x 480 where the fp32 matrix-core forms reach 6e-4, softmax weights in a tie of unrelated hypotheses within the stated 1e-4; 15-25 % slower (profiles/r05_k2_precise_ab.txt), every call that launches K2 honours it; bit 27 (0x8000000): RECORDS IN TWO PIECES -- the fast matrix-core form with the low parts of the pose records carried through twelve fp16 matrix-core instructions per 1 024 pairs chained onto the fp32 ones: measured: 85 % of the fast form's score error gone (0.112 -> 0.017 on scores of 1.5e5; softmax weights in a tie of unrelated hypotheses 4.4e-3 -> 7.2e-4) for +16 % of K2's time (profiles/r05_k2_reclo_ab.txt) -- a middle mode that does NOT reach the stated 1e-4 in ties (bit 25 does); k2_variant 80..83 select its register / occupancy trades; bit 26 (with bit 25, diagnostic): the precise form with ONLY its pose records rounded to float -- isolates what the fp32 record costs (it is 96 % of the fast forms' score error)
 *                 bit 28 (0x10000000): EXACT TRANSFORM (round 6) -- E = R.X + t from fixed-point fp16 pieces of the pose records and of the coordinates on the fp16
 *                 matrix core (one accumulation per row exact, the other below a few millimetres), the camera-frame point rounded to float ONCE, the distance
 *                 without a reciprocal of z (whose hardware approximation biases one hypothesis against another): e = n rsq(n z^2), n = (pu z - x)^2 + (pv z - y)^2
 *                 -- no cell above 1e-3 px over all cells of 256 x 640x480 (max 5.6e-4, mean 8.1e-6), softmax weights in a tie of unrelated hypotheses within
 *                 the stated 1e-4 (5.5e-5 .. 8.3e-5) at 1.03-1.05x the fast form's time in the same process (984-995 against 940-958 us; 0.64-0.65 of the HBM peak at the bench shape,
 *                 profiles/r06_k2_rsq_ab.txt; with reciprocal + Newton step + square root, k2_variant 84 / 93: 1 002-1 052; the precise mode: 1.85x).  Needs a map the vector kernels can read and a focal length
 *                 <= 1 024 px; coordinates beyond +-65.5 m take the fp32 transform chunk by chunk.  An arithmetic form that is ASKED for by bit 25 / 27 / 28 and
 *                 cannot run on the frame is an error (round 6), never a silent launch of another form.  k2_variant 84 / 85 / 89 / 93 / 94 / 95 are its tile and tail trades.
 *   "k2_exact_auto"  1 (default since round 6): the auto policy (k2_variant -1, none of the bits 1 / 25 / 26 / 27 set) launches the exact-transform form wherever
 *                 it applies and falls back to the fp32 matrix-core forms where it does not -- the library's default K2 holds every stated tolerance; 0: the
 *                 fp32 forms of rounds 2-5 (DSAC_K2_EXACT_AUTO in the environment of dsac_create)
 *   "k2_diag"     diagnostic switches of the precise form (bit 25): degrade ONE step at a time towards the fast forms' arithmetic (scripts/r06_k2_diag.py)
 *   "k6_walk_exact" 1: the scan of the two-launch refinement step (k_refine_walk, >= 32 problems on >= 16 384 cells) decides every cell by the reference's fp64
 *                 residual; 0 (default): an fp32 test with a proven error bound discards the cells that are certainly no inliers, the fp64 residual decides the
 *                 rest -- the same decisions bit for bit (A/B switch; tests/test_gpu_refine.py builds a map against the test)
 *   "k6_scan_tune" A/B of that scan, process-wide: problems per wave (0 = by the problem count, 1, 2, 4) | chunk cells << 8 (0 = auto, a multiple of 256) |
 *                 1 << 24 to switch the skip of chunks behind max_inl finished inliers off.  Results never depend on it
 *   "k6_waves"    waves per refinement problem of K6's inlier walk: 0 (default) = by the problem count (1 for a single problem -- the four-wave build costs the good-pose refinement of one image 2.6 us --, 4 up to 512 problems, 2 up to 1 024, else 1), 1 / 2 / 4 / 8
 *                 fixed.  Wave 0 runs the problem as before; when the first 256 cells of a step's permutation do not give max_inl inliers the walk goes on in
 *                 rounds of waves x 256 cells (counts meet in LDS, every wave compacts behind the inliers in front of its cells): the same list, inlier maps, step
 *                 counts and poses bit for bit; 128 whole-map walks on 640 x 480 15.3 -> 5.4 ms (profiles/r06_k6_walk_v2.txt).  With 0, calls of >= 32 problems on a
 *                 map of >= 16 384 cells (and no perturbed cells / fused loss) run every step as a scan of the step's cells + one LM launch instead (0.65 ms for the
 *                 same 128 walks, bit-identical again; "k6_walk_exact", "k6_scan_tune" above; profiles/r06_k6_walk.txt); a non-zero value keeps the fused kernel
 *   "refstream_mode"  see dsac_sample_refstream
 *   "k1_wpb", "k1_prio", "k1_hpw", "k1_minw"   K1 waves per workgroup (1), wave priority (3), hypotheses per wave (1), register budget in waves per SIMD (1)
 *   "k1_rl"       lanes per sampling attempt: 1 (default) = one lane per attempt, the quartic's roots in sequence, 64 attempts per round and
 *                 hypothesis; 4 = one lane per root, 16 attempts per round (the form the wpb / hpw / minw / share knobs below act on)
 *   "k1_wide"     waves per hypothesis of the one-lane-per-attempt form when there are few hypotheses: -1 auto (2 up to 512 hypotheses,
 *                 else 1), 0 = always 1, 2, 4 (4: up to 256 hypotheses) -- a round is 64 x waves attempts wide, the first accepted attempt is the same
 *   "k1_share"    4 or 8: the waves of a K1 workgroup evaluate the next attempts of their unfinished neighbours -- same first accepted attempt,
 *                 shorter tail; applied up to 1024 hypotheses (negative: always); 0 = off
 *   "k1_horn"     1 = align the P3P triangle with Horn's quaternion method by Jacobi sweeps, operation by operation as OpenCV's solvePnP(CV_P3P)
 *                 (K1 43 -> 230 us at 4096 hypotheses); 0 (default) = the same least-squares optimum in closed form (rounds 1-4: an orthonormal triad,
 *                 which is that optimum only when the P3P lengths are consistent -- on ill-conditioned minimal sets it is another pose, and another
 *                 pose accepts another set)
 *   "k1_cus"      > 0: the auxiliary stream of dsac_sample_ahead is created with a CU mask of that many CUs (before its first use)
 *   "device_args"  1: the caller promises that EVERY pointer argument from now on is a device pointer (dsac_device_alloc, torch, hipMalloc): the library skips
 *                 its per-argument hipPointerGetAttributes query (about a microsecond each, 20-odd per dsac_process_images).  A host pointer passed under
 *                 this promise is a caller bug with undefined behaviour; 0 (default): detect
 *   "seed_stride"  frame f of a frame batch draws from the random stream of seed + f * seed_stride (default 1).  Images sharded round-robin over W ranks
 *                 (rank r owns images r, r + W, ...) keep the seeds they have in the unsharded loop with seed_stride = W: results do not depend on W
 *   "tail_prio"   1: the deferred tails' streams are created with the highest stream priority (set before the first deferred dsac_process_images;
 *                   default 0 -- measured without effect on one GPU, profiles/r04_tail_prio.txt)
 *   "pi_defer_tail" 1: dsac_process_images defers its refinement tail, 2: its score tail (reduction, K3) as well (see dsac_join_tail);
 *                   0 (default): everything in stream order
 *   "k4_variant"  K4 main pass: -1 auto; 0 VALU form; 1 / 2 / 3 / 4 / 5 matrix-core form with 2 / 4 / 5 / 6 / 3 chunks per wave, 6 / 7 its high-occupancy builds (+ 10 x tile code + 100 x workgroups per CU)
 * The environment variables DSAC_K2_VARIANT, DSAC_K2_ORDER, DSAC_K2_FLAGS, DSAC_K1_WPB, DSAC_K1_PRIO, DSAC_K1_HPW, DSAC_K1_MINW, DSAC_K1_RL, DSAC_K1_WIDE, DSAC_K1_SHARE,
 * DSAC_K1_HORN, DSAC_K1_CUS, DSAC_K4_VARIANT, DSAC_TAIL_PRIO give the initial values at dsac_create. */
DSAC_API int dsac_set_option(dsac_ctx* ctx, const char* key, int value);

/* ---- device buffers for a host program that has no HIP toolchain --------------------------------------------- */
/* The reference keeps every operand in host memory (cv::Mat: jp::img_coord_t estObj core/types.h:43-51, the hypothesis vectors of
 * core/cnn_softam.h:971-988).  A C++ host that follows INTEGRATION.md keeps them in HBM instead and needs nothing but this header for it:
 * dsac_device_alloc / dsac_device_free hand out device memory of the context's GPU, dsac_host_alloc / dsac_host_free page-locked host
 * memory (copies to and from it are truly asynchronous), dsac_copy_async moves `bytes` between any two of them (or pageable host memory)
 * in the order of the context's stream and returns without waiting unless the runtime has to stage a pageable buffer.  Pointers from
 * dsac_device_alloc are "device pointers" wherever this header says so; buffers are not tied to the context beyond their GPU. */
DSAC_API int dsac_device_alloc(dsac_ctx* ctx, size_t bytes, void** out);
DSAC_API int dsac_device_free(dsac_ctx* ctx, void* p);
DSAC_API int dsac_host_alloc(dsac_ctx* ctx, size_t bytes, void** out);
DSAC_API int dsac_host_free(dsac_ctx* ctx, void* p);
DSAC_API int dsac_copy_async(dsac_ctx* ctx, void* dst, const void* src, size_t bytes);
DSAC_API int dsac_fill_zero_async(dsac_ctx* ctx, void* dst, size_t bytes);
/* Row i of dst = row rows[i] of src (device pointers, rows of row_bytes bytes, a multiple of 4), in the order of the context's stream, ONE launch for
 * up to 256 rows; `rows` is a HOST array that is read before the call returns.  This is how a training round draws its frames from a training set that
 * stays in HBM (core/train_ransac_softam.cpp:227-235 picks a random frame per round): n_rows coordinate maps -- and their sampling tables and ground
 * truths -- into the round's batch with three launches instead of 3 n_rows copies. */
DSAC_API int dsac_gather_rows(dsac_ctx* ctx, void* dst, const void* src, size_t row_bytes, int n_rows, const int32_t* rows);

/* ---- frame ------------------------------------------------------------------------------------ */
/* Replaces the (estObj, sampling, camMat) triple every reference function takes
 * (core/cnn_softam.h:319-323, 564-570, 663-671, 960-988). */
DSAC_API int dsac_set_frame(dsac_ctx* ctx, const float* xyz, const float* uv_or_null, int H, int W, float fx, float fy, float cx, float cy,
                   unsigned flags);

/* Frame batch: `frames` coordinate maps of the same H x W and camera, stored back to back (frame f at xyz + f*H*W*3); uv is one
 * shared H*W x 2 table (uv_per_frame = 0), one table per frame (uv_per_frame = 1) or NULL (implicit grid).  Independent frames
 * are the unit the path shards over (core/test_ransac_softam.cpp:97-230); batching them lets one launch carry several frames.
 * A batch is accepted by: dsac_score_hypotheses_frames, the pipelined pair dsac_sample_ahead / dsac_score_sampled, dsac_process_images, dsac_refine,
 * dsac_loss_frames, and -- since round 4 -- the backward calls dsac_dpnp, dsac_score_backward, dsac_soft_score_backward, dsac_last_pose_gradients,
 * dsac_refine_fd, dsac_backward_path1 and dsac_path1_and_softmax_backward: there N = frames x hypotheses per frame (per-hypothesis arrays frame-major),
 * every per-image argument (start / refined pose, ground truth, inlier map, J_hyp, obj_pixels, J_obj, n_obj, dL, v6) holds one slice per frame, grad_xyz
 * is frames x H*W x 3, and every stage is ONE launch over all frames; the results equal `frames` single-frame calls bit for bit
 * (core/train_ransac_softam.cpp:288-394 is one image per round; a batch is what the data-parallel step of SURVEY.md 5 puts on one GPU).  The score
 * backward is ONE launch for all frames when 16 | hypotheses per frame (up to 256 as one hypothesis tile per frame, beyond that -- round 6 -- as several equal tiles of
 * the same launch: 384 = 2 x 192, 512 = 2 x 256, their gradients meeting in fp64 atomics); with any other count, on maps its matrix-core form cannot read as vectors
 * (H*W or -- with the implicit grid -- W not a multiple of 4, buffers not 16-byte aligned), with the staged forms ("k4_variant" 0 or >= 1000) and in the
 * fp64 parity mode it runs frame by frame inside the call (F times the launches, the results of F single-frame calls).  Since round 5 the stages work on a batch one by one as well -- dsac_sample (sets drawn
 * here), dsac_reproject (128 | hypotheses per frame), dsac_softmax_frames, and the pair dsac_process_images_begin / dsac_process_images_finish, the
 * score-CNN seam of the batched fast path -- and so do the DSAC-variant calls (dsac_refine_all, dsac_refine_fd_sets, dsac_loss_batch, dsac_select_frames).
 * What still reports DSAC_ERR_INVALID while a batch is set: dsac_score_hypotheses (use dsac_score_hypotheses_frames), dsac_sample with GIVEN sets,
 * dsac_refine_fd_set (one hypothesis: use dsac_refine_fd_sets). */
DSAC_API int dsac_set_frames(dsac_ctx* ctx, int frames, const float* xyz, const float* uv_or_null, int uv_per_frame, int H, int W, float fx, float fy,
                    float cx, float cy, unsigned flags);
/* dsac_score_hypotheses for every frame of the batch in three launches (K1, K2, K3 over frames x hyps_per_frame hypotheses).
 * Frame f draws from the stream of seed + f, so the result equals `frames` single-frame calls with seeds seed, seed + 1, ...
 * hyps_per_frame must be a multiple of 128.  Outputs are frame-major: poses / sets_out / ok / scores / w [frames][hyps_per_frame],
 * err [frames*hyps_per_frame][H*W], entropy [frames], avg6 [frames][6].
 * With dsac_set_option("pi_defer_tail", 2) and device-resident arguments (dsac_score_hypotheses without given sets as well) the score tail (reduction of the per-tile sums, K3) runs on the tail stream
 * beside K1 of the NEXT call, which follows this call's K2 without a gap (the two small launches otherwise leave the chip idle: 12 of 946 us per
 * 16-frame step).  The contract is dsac_process_images' mode 2: everything but the error images is ordered on the context's stream only after
 * dsac_join_tail / another entry point / dsac_synchronize, and consecutive calls are given different arrays for poses / sets_out / ok / scores / w /
 * entropy / avg6 (the error images may share one buffer).  Results are the same bit for bit (tests/test_gpu_process_images.py). */
DSAC_API int dsac_score_hypotheses_frames(dsac_ctx* ctx, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clamp, float tau, float beta,
                                 double scale, double* poses, int32_t* sets_out, uint8_t* ok, float* err_or_null, double* scores_or_null, double* w,
                                 double* entropy_or_null, double* avg6_or_null);

/* ---- K1: minimal-set sampling + P3P ------------------------------------------------------------ */
/* Replaces the sampling loop of processImage, core/cnn_softam.h:1010-1060 (irand x4, alreadyChosen,
 * safeSolvePnP(CV_P3P) :1042, projectPoints :1046, 4-point re-projection check :1050-1059 with the
 * threshold truncated to int as at test_ransac_softam.cpp:51).  sets_or_null == NULL draws the sets with
 * the counter RNG above; otherwise the given N x 4 pixel indices are evaluated once each (this is also the
 * "re-solve P3P from the stored minimal set" step of dScore, core/cnn_softam.h:583-598).
 * Outputs: poses N x 6, sets_out N x 4 (pixel index y*W+x, the reference's imgIdx), ok N. */
DSAC_API int dsac_sample(dsac_ctx* ctx, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, double* poses,
                int32_t* sets_out, uint8_t* ok);

/* K1 in the REFERENCE'S OWN random stream (round 6).  The reference draws from ThreadRand (core/thread_rand.cpp:40-69): one std::mt19937(seed + t) per
 * OpenMP thread t, irand(0, n) = std::uniform_int_distribution<int>(0, n - 1) (:59-69, :95-98), x before y, a re-draw for a cell already in the set, a new
 * attempt after a failed P3P / re-projection check, no cap (core/cnn_softam.h:1010-1060); `#pragma omp parallel for` (static schedule) gives thread t the
 * hypotheses [t q + min(t, r), + q + (t < r)), q = N / threads, r = N % threads, which it serves in order from its stream.
 *   dsac_refstream_init     = ThreadRand::forceInit(seed) with omp_get_max_threads() == threads: the generators live in the context, like the reference's
 *                             static ones, and run on through successive images.
 *   dsac_refstream_discard  : generator `thread` skips n32 32-bit outputs -- what reference code OUTSIDE this path drew from it (stochasticSubSample,
 *                             core/cnn_softam.h:283-309: two drand = four outputs per cell, 6 400 for its 40 x 40 grid; irand(0, n) for the training frame: one).
 *   dsac_sample_refstream   : dsac_sample with the sets drawn from those generators.  The attempt sequence of a stream is a function of the stream alone,
 *                             so a window of attempts per stream is parsed, evaluated in parallel (K1's own P3P + check) and handed out in order; the
 *                             generators advance exactly as far as the sequential loop would have read.  max_attempts caps the attempts per stream (the
 *                             reference has none); hypotheses left unserved report ok = 0.  consumed32_or_null / attempts_or_null [threads]: outputs taken
 *                             from / attempts made on each stream by this call.  Synchronises the context's stream (a window that served too few is followed
 *                             by another).  One frame at a time.  "refstream_mode" (dsac_set_option): 0 = std::uniform_int_distribution as libstdc++ >= 11
 *                             computes it (Lemire's method), 1 = as libstdc++ <= 10 did (scaling + division), -1 = as the libstdc++ this library was built with.
 * Bit-identical minimal sets to the real reference's processImage on both golden frames, threads = 1 and 4 (tests/test_gpu_refstream.py). */
DSAC_API int dsac_refstream_init(dsac_ctx* ctx, unsigned seed, int threads);
DSAC_API int dsac_refstream_discard(dsac_ctx* ctx, int thread, unsigned long long n32);
DSAC_API int dsac_sample_refstream(dsac_ctx* ctx, int N, float thr, long long max_attempts, double* poses, int32_t* sets_out, uint8_t* ok,
                                   unsigned long long* consumed32_or_null, long long* attempts_or_null);

/* ---- K2: batched reprojection -> error images and/or soft-inlier scores -------------------------- */
/* Replaces the N getDiffMap calls of core/cnn_softam.h:1067-1069 (getDiffMap :319-362):
 * err[h][p] = min(|uv_p - project(K, pose_h, xyz_p)|, clamp)  (clamp = CNN_OBJ_MAXINPUT = 100, lua_calls.h:36).
 * soft[h] = sum_p sigmoid(beta * (tau - err[h][p]))  is the DSAC++-style soft-inlier count named by
 * north_star (not in the reference).  Either output may be NULL. */
DSAC_API int dsac_reproject(dsac_ctx* ctx, int N, const double* poses, float clamp, float* err_or_null, float tau, float beta,
                   double* soft_or_null);

/* ---- K3: softmax / entropy / soft-argmax pose --------------------------------------------------- */
/* Replaces softMax core/cnn_softam.h:535-553, entropy :80-88 and the weighted pose average :1082-1094.
 * w = softmax(scale * scores); entropy in bits; avg6 = sum_h w_h * poses[h].  entropy/avg6/poses may be NULL. */
DSAC_API int dsac_softmax(dsac_ctx* ctx, int N, const double* scores, double scale, double* w, double* entropy_or_null,
                 const double* poses_or_null, double* avg6_or_null);
/* The same for `frames` independent groups of hyps_per_frame consecutive scores in ONE launch (a workgroup per frame): scores / w / poses
 * [frames][hyps_per_frame], entropy [frames], avg6 [frames][6].  Needs no frame to be set -- K3 reads scores and poses only.  This is the K3 of a
 * frame batch whose scores come from a score model outside the library (core/cnn_softam.h:1072-1078: forward(diffMaps) -> softMax). */
DSAC_API int dsac_softmax_frames(dsac_ctx* ctx, int frames, int hyps_per_frame, const double* scores, double scale, double* w, double* entropy_or_null,
                        const double* poses_or_null, double* avg6_or_null);

/* ---- K1 + K2 + K3 in one call: the hypothesis-scoring half of processImage ------------------------ */
/* Replaces core/cnn_softam.h:1010-1094 with the soft-inlier score in the place of the score CNN (:1072):
 * sample N hypotheses (dsac_sample), reproject (dsac_reproject: error images optional, soft-inlier sums always),
 * w = softmax(scale * soft), entropy, soft-argmax pose.  Same results as the three separate calls; the fused
 * form saves the pose-staging launch and the host round trips.  scores_or_null receives the soft-inlier sums. */
DSAC_API int dsac_score_hypotheses(dsac_ctx* ctx, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, float clamp, float tau,
                          float beta, double scale, double* poses, int32_t* sets_out, uint8_t* ok, float* err_or_null, double* scores_or_null,
                          double* w, double* entropy_or_null, double* avg6_or_null);

/* The same work as a two-slot software pipeline inside one context: dsac_sample_ahead(slot) enqueues K1 for a LATER
 * frame on an auxiliary stream; dsac_score_sampled(slot) enqueues K2 -> K3 for that slot on the context's stream.
 * Called as  ahead(0) ; { ahead(1-s) ; score(s) ; s = 1-s } ...  the latency-bound sampling of frame i+1 runs underneath
 * the bandwidth-bound scoring of frame i, the small K3 tail of frame i runs under K2 of frame i+1, and K2 launches follow
 * each other back to back on the context's stream.  The error images are complete in stream order; scores / w / entropy /
 * avg6 are complete after dsac_synchronize (they are produced on the auxiliary stream).  Device pointers only
 * (the calls never block); poses/sets_out/ok of a slot must stay untouched until its score call has been issued
 * (the library orders the slot's next dsac_sample_ahead behind the readers of its previous use), and dsac_synchronize
 * waits for all streams.  A slot remembers the frame that was current when it was sampled and is scored against that frame, so
 * a stream of different frames is pipelined by calling dsac_set_frame(..., DSAC_FRAME_BORROW) before each dsac_sample_ahead;
 * frames the library copies itself (no DSAC_FRAME_BORROW) cannot be replaced while a slot is sampled but not yet scored
 * (DSAC_ERR_INVALID).  Each slot alternates strictly: sample_ahead, score_sampled, sample_ahead, ... */
DSAC_API int dsac_sample_ahead(dsac_ctx* ctx, int slot, int N, uint64_t seed, const int32_t* sets_or_null, float thr, int max_tries, double* poses,
                      int32_t* sets_out, uint8_t* ok);
DSAC_API int dsac_score_sampled(dsac_ctx* ctx, int slot, float clamp, float tau, float beta, double scale, const double* poses, float* err_or_null,
                       double* scores, double* w, double* entropy_or_null, double* avg6_or_null);

/* ---- K5: dPNP ------------------------------------------------------------------------------------ */
/* Replaces dPNP core/cnn_softam.h:101-146 for the minimal (4-point, CV_P3P) case: central differences
 * (float eps, sequential float perturbation of the object points) of the jp 6-vector of the P3P pose.
 * J is N x 6 x 12; all-zero for a hypothesis whose differences contain NaN (:141-142). */
DSAC_API int dsac_dpnp(dsac_ctx* ctx, int N, const int32_t* sets, float eps, double* J);

/* ---- K4: score backward -------------------------------------------------------------------------- */
/* Replaces dScore part (iii), core/cnn_softam.h:609-645, plus the sum over hypotheses at
 * core/train_ransac_softam.cpp:382-383:  grad_xyz[p] += sum_h ( d_err[h][p] * dProjectdObj(h,p) )
 * and, for the 4 support pixels of h, += (sum_p d_err[h][p] * dProjectdHyp(h,p)) * dPNP(h).
 * poses are the cv poses of the hypotheses (as re-solved at :597-598); dpnp N x 72 or NULL (computed
 * internally with eps = 0.1f).  grad_xyz is H*W x 3 doubles, ACCUMULATED into.
 * Determinism: with one hypothesis tile (N <= 256) or a frame batch the main pass adds into grad_xyz with hardware fp64 atomics -- a cell receives at most
 * two such additions per call (a pixel tile split between two workgroups) on top of the value it held, so two runs may differ in the last bit of a cell;
 * dsac_set_option("k4_variant", 1000 + v) (1999: automatic form) is the staged, bit-reproducible form (fp32 partial sums + a reduction launch; on a frame batch
 * it runs frame by frame).  The atomics need ordinary device memory: a managed / fine-grained grad_xyz takes the staged form by itself, and is refused on a frame
 * batch; under "device_args" = 1 the caller's promise includes that. */
DSAC_API int dsac_score_backward(dsac_ctx* ctx, int N, const double* poses, const int32_t* sets, const float* d_err, const double* dpnp_or_null,
                        unsigned flags, double* grad_xyz);
/* The backward calls need fx == fy: the reference's Jacobians use the single focal length camMat(0,0) for both axes
 * (core/cnn_softam.h:406,466); a camera with two focal lengths is rejected with DSAC_ERR_INVALID rather than differentiated
 * inconsistently with the forward kernels.  Quirk 7 of the reference (dProjectdHyp writes the re-derived rotation back into the
 * hypothesis through a const reference, :506-508, so the rotation drifts by round-off from pixel to pixel): the fast fp32 kernels are
 * the "fixed" mode (rotation re-derived once per hypothesis); dsac_score_backward reproduces the write-back with
 * DSAC_BWD_PARITY_FP64 | DSAC_BWD_QUIRK_ROT_WRITEBACK (fp64, the reference's evaluation order; 6e-12 against the oracle's
 * quirk_rot_writeback mode), and tests/test_gpu_backward.py::test_quirk7_rot_writeback bounds the difference between the two modes.
 * Same with the soft-inlier score: d_err[h][p] = g[h] * d soft[h] / d err[h][p], formed in-kernel
 * (no N x P read).  g = dLoss/d soft[h]. */
DSAC_API int dsac_soft_score_backward(dsac_ctx* ctx, int N, const double* poses, const int32_t* sets, const double* g, float clamp, float tau,
                             float beta, const double* dpnp_or_null, unsigned flags, double* grad_xyz);

/* The soft-inlier score as an EXTERNAL score model, backward half: the gradient images a score model hands to dsac_score_backward
 * (core/train_ransac_softam.cpp:378-383: backward -> dScore), written out for the soft-inlier score -- d_err[h][p] = g[h] * (-beta) s (1 - s),
 * s = sigmoid(beta (tau - err[h][p])), 0 where err sits on the clamp; err / d_err N x H*W float32 (16-byte aligned, 4 | H*W), g N doubles.
 * With dsac_process_images_begin's `soft` output as the forward half, a host without device code of its own (the C++ programs of dsac_amd/host)
 * drives the whole score-CNN seam -- error images out, scores in, score gradients out, gradient images in -- and must reproduce the built-in
 * dsac_soft_score_backward to fp32 rounding (tests/test_gpu_seam.py, train_ransac_softam -seam 1). */
DSAC_API int dsac_soft_score_derr(dsac_ctx* ctx, int N, const double* g, const float* err, float clamp, float tau, float beta, float* d_err);

/* The per-hypothesis 1 x 6 pose gradients of the most recent dsac_score_backward / dsac_soft_score_backward call on
 * this context: G6[h] = sum over cells of d_err[h][p] * dProjectdHyp(p) (the accumulation of core/cnn_softam.h:631-632
 * before its product with dPNP; columns = jp Rodrigues vector, translation in mm).  N must not exceed that call's N. */
DSAC_API int dsac_last_pose_gradients(dsac_ctx* ctx, int N, double* G6);

/* ---- K6: inlier refinement (LM-PnP) and its finite-difference Jacobians -------------------------- */
/* Replaces the refinement loop of processImage core/cnn_softam.h:1099-1154 (B = 1, fills inlier_map) and
 * the replay helper refine() :663-723 (B replicas).  perm is steps x H*W pixel indices (the reference's
 * pixelIdxs); per step the first max_inl cells along perm with err < thr are collected, the loop stops if
 * fewer than min_inl (50), otherwise solvePnP(CV_ITERATIVE, useExtrinsicGuess) restarts from the current
 * pose.  pert_px_c (B x 2: pixel or -1, channel) / pert_value (B) replace one coordinate per replica, which
 * is dRefineObj's localEstObj (:887,901).  out_poses B x 6 (cv).  inlier_map (H*W int32, += 1 per
 * selection) is only written for replica 0 and only when non-NULL.  With a frame batch (dsac_set_frames) B = frames x k problems, problem b
 * refines against frame b / k, no perturbations, and inlier_map (if given) is B x H*W, one map per problem. */
DSAC_API int dsac_refine(dsac_ctx* ctx, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                const int32_t* pert_px_c_or_null, const float* pert_value_or_null, double* out_poses, int32_t* inlier_map_or_null,
                int32_t* steps_done_or_null);
/* Replaces dRefineHyp core/cnn_softam.h:738-836 (J_hyp 6 x 6; eps_hyp = 0.001f) and dRefineObj :853-923
 * (eps_obj = 2.f, every skip = (int)(1/sub_sample)-th cell of inlier_map > 0 in x-outer/y-inner order,
 * scaled by skip).  All 12 + 6*n_obj replicas run as one batch.  dRefineObj's 6 x 3P matrix is returned
 * sparse: obj_pixels[i] = pixel index, J_obj[i] = 6 x 3 block, i < *n_obj <= cap. */
DSAC_API int dsac_refine_fd(dsac_ctx* ctx, const double* init_pose, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                   const int32_t* inlier_map, float sub_sample, float eps_hyp, float eps_obj, double* J_hyp, int32_t* obj_pixels,
                   double* J_obj, int cap, int32_t* n_obj);

/* ---- DSAC variant (core/cnn.h, the probabilistic-selection twin of the soft-argmax path) ------------------------ */
/* All N hypotheses refined as one batch: the per-hypothesis loop of processImage core/cnn.h:1155-1215.  perm is the
 * shared steps x H*W permutation (every hypothesis re-seeds the same default std::mt19937, :1169).  inlier_maps
 * (N x H*W int32, zeroed here) receives one hit-count map per hypothesis; with sets (N x 4) the cells of each
 * hypothesis' own minimal set are cleared afterwards (:1208-1214). */
/* Frame batch (round 5; core/cnn.h:1154-1230 for F images at once -- SURVEY.md 8(f)1: "the natural way to fill the GPU"): N = frames x hypotheses per
 * frame, hypothesis h refines against frame h / (N / frames); init_poses / sets / out_poses / steps_done frame-major, inlier_maps N x H*W.  ONE launch of
 * N waves; equals `frames` single-frame calls bit for bit. */
DSAC_API int dsac_refine_all(dsac_ctx* ctx, int N, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                    const int32_t* sets_or_null, double* out_poses, int32_t* inlier_maps_or_null, int32_t* steps_done_or_null);
/* Replaces dRefine core/cnn.h:854-990 for ONE hypothesis given by its minimal set: the refinement restarts from P3P of
 * the set (:797-800), so besides dRefineObj's inlier cells the first three set points are perturbed as well
 * (+-eps_obj = 2.f on the map and on the P3P input alike).  J_set is 6 x 9 (columns pt*3 + c, pt < 3), J_obj / obj_pixels /
 * n_obj as in dsac_refine_fd (scaled by skip; J_set is not, :923). */
DSAC_API int dsac_refine_fd_set(dsac_ctx* ctx, const int32_t* set4, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                       const int32_t* inlier_map, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj, int cap,
                       int32_t* n_obj);
/* dsac_refine_fd_set for M hypotheses in ONE batch of M * (18 + 6*cap) refinement problems -- the loop over hypotheses of
 * core/train_ransac.cpp:314-339 (the reference runs it under OpenMP).  sets M x 4, inlier_maps M x H*W; outputs J_set M x 6 x 9,
 * obj_pixels M x cap, J_obj M x cap x 6 x 3, n_obj M (entries beyond n_obj[m] are untouched). */
DSAC_API int dsac_refine_fd_sets(dsac_ctx* ctx, int M, const int32_t* sets, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                        const int32_t* inlier_maps, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj, int cap,
                        int32_t* n_obj);
/* Frame batch: with dsac_refine_fd_sets M must be frames x (hypotheses per frame), hypothesis m lives in frame m / (M / frames).  The trainer differentiates
 * only the hypotheses that carry weight (core/train_ransac.cpp:318: probability > 1e-4), a different number per image: dsac_refine_fd_sets_frames takes
 * frame_of[m] (M int32, host or device) -- the frame hypothesis m was sampled from -- and runs all M * (18 + 6*cap) replicas of all images in one launch. */
DSAC_API int dsac_refine_fd_sets_frames(dsac_ctx* ctx, int M, const int32_t* sets, const int32_t* frame_of, const int32_t* perm, int steps, int max_inl, int min_inl,
                               float thr, const int32_t* inlier_maps, float sub_sample, float eps_obj, double* J_set, int32_t* obj_pixels, double* J_obj,
                               int cap, int32_t* n_obj);
/* maxLoss / dLossMax for B estimates against one ground truth: the losses[] of expectedMaxLoss core/cnn.h:137-150 and
 * the per-hypothesis dLossMax of core/train_ransac.cpp:345-349.  out4 is B x 4, J6 B x 6 (layouts of dsac_loss). */
DSAC_API int dsac_loss_batch(dsac_ctx* ctx, int B, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null);
/* The same for a frame batch: frames x per_frame estimates (frame-major), estimate b against the ground truth of frame b / per_frame (gt_jp6 frames x 6). */
DSAC_API int dsac_loss_batch_frames(dsac_ctx* ctx, int frames, int per_frame, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null);
/* DSAC variant, the three small reductions over the N hypotheses of an image on the device (one launch, no host round trip):
 *   hyp_idx        = draw(probs)                 core/cnn.h:102-127: entries below EPS = 1e-8 skipped, the entry whose cumulative probability
 *                                                first exceeds u * sum (u in [0, 1) supplied by the caller in place of drand); u < 0: the most
 *                                                probable entry (randomDraw = false)
 *   expected_loss  = sum_i probs[i] losses[i]    expectedMaxLoss core/cnn.h:137-150 (index order)
 *   score_gradients[i] = probs[i] losses[i] - sum_j probs[i] probs[j] losses[j]      dSMScore core/cnn.h:737-742 (term by term, index order)
 * losses[i * loss_stride]: loss_stride = 4 reads the first column of dsac_loss_batch's out4 in place. */
DSAC_API int dsac_select(dsac_ctx* ctx, int N, const double* probs, const double* losses, int loss_stride, double u, int32_t* hyp_idx_or_null,
                double* expected_loss_or_null, double* score_gradients_or_null);
/* ... for `frames` images in one launch (a workgroup per image): probs / losses / score_gradients [frames][N], u / hyp_idx / expected_loss [frames]. */
DSAC_API int dsac_select_frames(dsac_ctx* ctx, int frames, int N, const double* probs, const double* losses, int loss_stride, const double* u,
                       int32_t* hyp_idx_or_null, double* expected_loss_or_null, double* score_gradients_or_null);

/* ---- producer side: patch gather for the scene-coordinate CNN ----------------------------------------------------- */
/* Replaces the patch assembly of getCoordImg core/cnn_softam.h:224-254 in the table layout of pushMaps core/lua_calls.h:63-80:
 * patches[i][c][y][x] = (float) bgr[(sy + y) * W + (sx + x)][c] with (sx, sy) = sampling_xy[i] - patch/2.  bgr is the H x W x 3
 * uint8 image (jp::img_bgr_t), sampling_xy n x (x, y) int32 (the reference's `sampling`), patch = CNN_RGB_PATCHSIZE (42).
 * A window that leaves the image (the reference skips such patches, :235-239) is written as zeros and counted in *skipped. */
DSAC_API int dsac_gather_patches(dsac_ctx* ctx, const uint8_t* bgr, int H, int W, const int32_t* sampling_xy, int n, int patch, float* patches,
                        int32_t* skipped_or_null);

/* ---- K7: pose loss ---------------------------------------------------------------------------------- */
/* Replaces maxLoss core/maxloss.h:69-79 (+ getInvHyp :39-61, Hypothesis::calcAngularDistance
 * Hypothesis.cpp:137-143) and dLossMax :87-198.  est is a cv pose (converted with cv2our, types.h:186-214,
 * as at cnn_softam.h:1160-1163 / train_ransac_softam.cpp:301-304); gt_jp6 is the ground truth as the jp
 * 6-vector poseGT.getRodVecAndTrans().  out4 = {loss, rotErr[deg], tErr[mm], correct(5deg/50mm)};
 * J6_or_null = dLossMax w.r.t. the jp 6-vector of est. */
DSAC_API int dsac_loss(dsac_ctx* ctx, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null);
/* maxLoss / dLossMax of B estimates, each against ITS OWN ground truth (est_cv6, gt_jp6 B x 6): the per-image evaluation of
 * core/test_ransac_softam.cpp:129-157 for a batch of images. */
DSAC_API int dsac_loss_frames(dsac_ctx* ctx, int B, const double* est_cv6, const double* gt_jp6, double* out4, double* J6_or_null);

/* ---- the whole test-time unit of work ------------------------------------------------------------------ */
/* processImage (core/cnn_softam.h:960-1179, called per image by core/test_ransac_softam.cpp:97-157) for EVERY frame set with dsac_set_frame /
 * dsac_set_frames, one launch per stage for all frames: K1 sample + P3P, K2 error images + soft-inlier sums (the score-CNN seam; scores are
 * scale * soft-inlier count), K3 softmax / entropy / soft-argmax pose, K6 the refinement loop (one wave per frame; perm = steps x H*W pixel
 * permutations shared by all frames -- the reference re-seeds a default mt19937 per image, :1104), K7 maxLoss against each frame's ground truth
 * (gt_jp6 frames x 6; NULL with out4 NULL: no loss).  Frame f draws from the random stream of seed + f, i.e. the result equals F single-frame
 * calls with seeds seed, seed + 1, ... bit for bit.  Outputs: poses / sets_out / ok / w (frames * hyps_per_frame), entropy (frames),
 * avg6 / ref6 (frames x 6, cv), steps_done (frames), inlier_maps (frames x H*W, zeroed here, or NULL), out4 (frames x 4: loss, rotErr deg,
 * tErr mm, correct), err_or_null (frames * hyps_per_frame x H*W), scores_or_null. */
DSAC_API int dsac_process_images(dsac_ctx* ctx, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clamp, float tau, float beta, double scale,
                        const int32_t* perm, int steps, int max_inl, int min_inl, const double* gt_jp6_or_null, double* poses, int32_t* sets_out,
                        uint8_t* ok, float* err_or_null, double* scores_or_null, double* w, double* entropy, double* avg6, double* ref6,
                        int32_t* steps_done, int32_t* inlier_maps_or_null, double* out4_or_null);
/* With dsac_set_option("pi_defer_tail", 1) and device-resident arguments, dsac_process_images hands its refinement tail (K6, K7: a latency chain on
 * one wave per frame) to a stream of its own, where it runs under sampling and scoring of the NEXT dsac_process_images call -- the loop over image
 * batches of core/test_ransac_softam.cpp:97-230 pipelined by one batch.  The tail's outputs (ref6, steps_done, inlier_maps, out4) are then ordered
 * on the context's stream only after dsac_join_tail (enqueues the dependency, does not block the host), any other entry point that enqueues work, or
 * dsac_synchronize; everything else (poses, sets, ok, err, scores, w, entropy, avg6) is in stream order as always.  A following dsac_process_images
 * must therefore be given other ref6 / steps_done / inlier_maps / out4 buffers if the previous batch's have not been consumed yet.
 * The tail also READS after the call has returned: the frame's xyz / uv (a DSAC_FRAME_BORROW frame is the caller's memory), perm, gt_jp6 and the avg6
 * it starts from.  None of them may be overwritten -- not even by work enqueued on the context's stream, which is not ordered against the tail --
 * before dsac_join_tail, another entry point that enqueues work, or dsac_synchronize; a following dsac_process_images may use the same perm / gt and
 * other frames, and orders its own write of avg6 behind the previous tail.  A pipeline that refills ONE borrowed coordinate buffer batch after batch
 * must call dsac_join_tail before the refill (tests/test_gpu_process_images.py::test_deferred_tail_and_a_reused_borrowed_frame_buffer).
 * dsac_set_option("pi_defer_tail", 2) moves the score tail to that stream as well: the reduction of the per-tile soft-inlier sums and K3 of batch i run
 * beside K1 of batch i + 1, which then follows K2 of batch i without a gap (the reduction and K3 are two small launches that leave the chip idle
 * while they run in order).  Then EVERY output of the call except the error images -- poses, sets_out, ok, scores, w, entropy, avg6 and the tail's
 * four -- is ordered on the context's stream only after dsac_join_tail / another entry point / dsac_synchronize, and consecutive calls must be given
 * different arrays for all of them; the call after the next may reuse them (the library orders its K1 behind K3 of the call two back, and its tail
 * behind that call's tail).  Because consecutive tails are independent then, SMALL calls (up to two 640 x 480 images x 256 hypotheses' worth of
 * pairs) alternate between two tail streams: in a loop of single images -- the loop of core/test_ransac_softam.cpp:97 as it stands -- the one-wave
 * refinement chains of images i and i + 1 run side by side and the loop is bound by K1 + K2: 173 us per image in order, 122 with mode 1, 81 with
 * mode 2 (profiles/r04_two_tails_ab.txt).  Results are the same bit for bit in all three modes (tests/test_gpu_process_images.py). */
DSAC_API int dsac_join_tail(dsac_ctx* ctx);

/* ---- the score-CNN seam of the batched fast path ------------------------------------------------------------------------------------------
 * dsac_process_images cut where the reference calls its score CNN (core/cnn_softam.h:1066-1078: getDiffMap x N -> forward(diffMaps) -> softMax;
 * core/lua_calls.h:89-105 pushes the N error images to Lua number by number).  For every frame set with dsac_set_frame / dsac_set_frames:
 *   dsac_process_images_begin   K1 sample + P3P, K2 -> err [frames * hyps_per_frame][H*W] in the caller's HBM buffer, the order the reference hands the
 *                               maps to the CNN (and, if soft is non-NULL, the soft-inlier sums as well); poses / sets_out / ok as dsac_process_images.
 *   ... the caller's score model turns err into scores [frames * hyps_per_frame] (doubles) -- on the context's stream (dsac_set_stream adopts torch's),
 *       or with its own ordering against it ...
 *   dsac_process_images_finish  K3 per frame on scale * scores (softmax / entropy / soft-argmax pose of `poses`), K6 the refinement of every frame, K7 the
 *                               loss -- arguments and outputs as the second half of dsac_process_images.
 * Frame f draws from the random stream of seed + f * seed_stride; scores equal to the soft-inlier sums (begin's `soft`) and scale = the score scale give
 * the results of dsac_process_images bit for bit (tests/test_gpu_seam.py).  "pi_defer_tail" applies to the pair: 1 = K6 / K7 run on the tail stream under
 * the NEXT begin's K1 / K2; 2 = K3 as well (it starts behind an event the finish call records on the context's stream, i.e. behind the score model);
 * ordering of the outputs, dsac_join_tail / dsac_tail_wait and the rule that consecutive calls are given different arrays in mode 2 are those of
 * dsac_process_images.  One pair at a time per context: finish must follow its begin with the same frames and hyps_per_frame (another
 * dsac_process_images or begin abandons an open begin).  The backward half of the seam is dsac_score_backward with the score model's d_err
 * (core/train_ransac_softam.cpp:378-383) -- on a frame batch since round 4. */
DSAC_API int dsac_process_images_begin(dsac_ctx* ctx, int hyps_per_frame, uint64_t seed, float thr, int max_tries, float clamp, float tau, float beta,
                              double* poses, int32_t* sets_out, uint8_t* ok, float* err, double* soft_or_null);
DSAC_API int dsac_process_images_finish(dsac_ctx* ctx, int hyps_per_frame, const double* scores, double scale, const int32_t* perm, int steps, int max_inl,
                               int min_inl, float thr, const double* gt_jp6_or_null, const double* poses, double* w, double* entropy, double* avg6,
                               double* ref6, int32_t* steps_done, int32_t* inlier_maps_or_null, double* out4_or_null);
/* The same dependency for ANOTHER stream: `hip_stream` (a hipStream_t of the context's device) waits for the deferred tail that is in flight -- and
 * thereby for the dsac_process_images call it belongs to and everything the context's stream held before that call (the tail starts behind that call's
 * K3); the context's own stream is not held up, nothing is inserted into it, and the tail stays pending for it.  This is how a consumer of the tail's
 * outputs (a copy to the host, the result gather of a multi-GPU evaluation) runs beside sampling / scoring of the next batch instead of in front of
 * it.  Without a tail in flight it is an ordinary cross-stream dependency on the context's stream as it is now (one event record). */
DSAC_API int dsac_tail_wait(dsac_ctx* ctx, void* hip_stream);

/* ---- gradient assembly ------------------------------------------------------------------------------ */
/* Replaces core/train_ransac_softam.cpp:344-376: with v6 = dLoss/dRef * dRef/dAvg (1 x 6),
 * grad_xyz[support px of h] += v6 * w_h * dPNP_h  (path I, second term) and the softmax backward
 * g_j = w_j * (F_j - sum_h w_h F_h),  F_h = v6 . [rvec_h ; tvec_h / 1000]  (written O(N^2) in the reference). */
DSAC_API int dsac_path1_and_softmax_backward(dsac_ctx* ctx, int N, const double* v6, const double* w, const double* poses, const int32_t* sets,
                                    const double* dpnp, double* grad_xyz, double* g);

/* The whole path-I half of the trainer's backward section, core/train_ransac_softam.cpp:294-376, enqueued as one chain with no host round
 * trip: dLossMax at the refined pose (ref_cv6 vs gt_jp6) -> dRefineObj / dRefineHyp of the refinement that started at avg_cv6 (one batch
 * of 12 + 6n finite-difference replicas; perm / steps / max_inl / min_inl / thr / inlier_map / sub_sample as dsac_refine_fd, eps_hyp 0.001f,
 * eps_obj 2.f in the reference) -> grad_xyz[inlier cells] += dL . dRefineObj and v6 = dL . dRefineHyp -> dPNP of the N minimal sets ->
 * grad_xyz[support cells of h] += v6 . w_h dPNP_h and the softmax backward g_j = g_scale * w_j (F_j - sum_h w_h F_h).  g (N) are the score
 * gradients path II continues from (the score CNN's backward, or dsac_soft_score_backward with g_scale = the score scale alpha);
 * dpnp_out_or_null (N x 72) can be handed on to dsac_score_backward.  grad_xyz (H*W x 3) is ACCUMULATED into.  dL_out / v6_out: the 1 x 6
 * dLoss/dRef and dLoss/dAvg for logging.  The reference computes both refinement Jacobians unconditionally; so does this call. */
DSAC_API int dsac_backward_path1(dsac_ctx* ctx, int N, const double* poses, const int32_t* sets, const double* w, const double* avg_cv6, const double* ref_cv6,
                        const double* gt_jp6, const int32_t* perm, int steps, int max_inl, int min_inl, float thr, const int32_t* inlier_map, float sub_sample,
                        float eps_hyp, float eps_obj, double g_scale, double* dpnp_out_or_null, double* grad_xyz, double* g, double* dL_out_or_null,
                        double* v6_out_or_null);

/* ---- stream scheduling ---------------------------------------------------------------------------------- */
/* Optional gate around the one bandwidth-bound kernel (K2): before launching it the context's stream waits for
 * `wait_before` (a hipEvent_t; a never-recorded event does not block), after it `record_after` is recorded.
 * Two contexts working on alternate frames cross-wire their events so that their K2 launches run back to back
 * -- never competing for HBM -- while the latency-bound kernels (K1 sampling, K3) of one frame fill the bubbles of
 * the other.  NULL removes the gate. */
DSAC_API int dsac_set_k2_events(dsac_ctx* ctx, void* wait_before_or_null, void* record_after_or_null);

/* ---- measurement hooks (bench.py's roofline leg) ----------------------------------------------------- */
/* When enabled, a hipEvent pair is recorded on the context's stream immediately around every launch of the
 * dominant kernel (K2 k_reproject, and K4 k_score_backward).  dsac_profile_read waits for the recorded
 * events and returns the summed kernel time in ms and the launch count per kernel (which = 0: K2, 1: K4). */
DSAC_API int dsac_profile_enable(dsac_ctx* ctx, int on);
DSAC_API int dsac_profile_read(dsac_ctx* ctx, int which, double* ms_total, int* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* DSAC_HIP_H */

// k_sample.hip -- K1 (minimal-set sampling + P3P) and K5 (dPNP) of the gfx950 DSAC engine.
//
// K1 replaces the rejection loop of processImage (core/cnn_softam.h:1010-1060).  The reference runs one
// OpenMP thread per hypothesis and retries sequentially; here ONE WAVE owns a hypothesis and evaluates 64
// consecutive attempts at once, one lane per attempt with the quartic's roots in sequence (the default since
// round 2; the first form -- 16 attempts, 4 lanes per attempt, one per root -- is kept behind k1_rl = 4), each
// attempt with its own counter-based draws.  A ballot picks the lowest accepted attempt, which is exactly the
// attempt the sequential loop would have stopped at, so the result does not depend on the wave width or on
// scheduling.  P3P runs in fp64 in registers (dmath.h).
//
// K5 replaces dPNP (core/cnn_softam.h:101-146): four lanes (one per quartic root) per (hypothesis, coordinate, +/-) P3P solve,
// 96 lanes = one two-wave workgroup per hypothesis; central differences are formed after a wave-local exchange.
#include "kernels.h"
#include "dmath.h"
#include "refstream.h"

namespace dk {

DM_INLINE dm::Cam make_cam(const FrameDev& F) { return dm::Cam{(double)F.fx, (double)F.fy, (double)F.cx, (double)F.cy}; }

// The four points of a minimal set: all coordinate loads first, then (one uniform branch) all position loads.  Written point by point
// with the position branch inside, the compiler waits for every point before it issues the next: four memory round trips
// per attempt instead of one -- about half of a K1 round.
DM_INLINE void load_set(const FrameDev& F, const int32_t set4[4], float X[4][3], float uv[4][2]) {
    int p[4];
#pragma unroll
    for (int j = 0; j < 4; j++) p[j] = min(max(set4[j], 0), F.P - 1);
#pragma unroll
    for (int j = 0; j < 4; j++) { X[j][0] = F.xyz[(size_t)p[j] * 3]; X[j][1] = F.xyz[(size_t)p[j] * 3 + 1]; X[j][2] = F.xyz[(size_t)p[j] * 3 + 2]; }
    if (F.uv) {
#pragma unroll
        for (int j = 0; j < 4; j++) { uv[j][0] = F.uv[(size_t)p[j] * 2]; uv[j][1] = F.uv[(size_t)p[j] * 2 + 1]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { const int y = p[j] / F.W; uv[j][0] = (float)(p[j] - y * F.W); uv[j][1] = (float)y; }
    }
}

// P3P + the 4-point re-projection check of core/cnn_softam.h:1042-1059.
template <bool HORN>
DM_INLINE bool solve_and_check(const FrameDev& F, const int32_t set4[4], int thr_int, double cv6[6]) {
    float X[4][3], uv[4][2];
    load_set(F, set4, X, uv);
    const dm::Cam K = make_cam(F);
    if (!dm::p3p<HORN>(X, uv, K, cv6)) return false;
    double R[9];
    dm::rodrigues_v2m<false>(cv6, R, nullptr);
    bool good = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float u, v;
        dm::project_f(R, cv6 + 3, K, X[j][0], X[j][1], X[j][2], u, v);
        const float dx = uv[j][0] - u, dy = uv[j][1] - v;
        good = good && (sqrt((double)dx * dx + (double)dy * dy) < (double)thr_int);
    }
    return good;
}

// Staged K2 record of a cv pose (same arithmetic as k_pose_prep in k_forward.hip: fp64 Rodrigues, intrinsics folded in).
DM_INLINE void write_staged_R(const FrameDev& F, const double R[9], const double cv6[6], float* o) {
    const double dfx = F.fx, dfy = F.fy;
    o[0] = (float)(dfx * R[0]); o[1] = (float)(dfx * R[1]); o[2] = (float)(dfx * R[2]); o[3] = (float)(dfx * cv6[3]);
    o[4] = (float)(dfy * R[3]); o[5] = (float)(dfy * R[4]); o[6] = (float)(dfy * R[5]); o[7] = (float)(dfy * cv6[4]);
    o[8] = (float)R[6]; o[9] = (float)R[7]; o[10] = (float)R[8]; o[11] = (float)cv6[5];
}
DM_INLINE void write_staged(const FrameDev& F, const double cv6[6], float* o) {
    double R[9];
    dm::rodrigues_v2m<false>(cv6, R, nullptr);
    write_staged_R(F, R, cv6, o);
}

// Draw the minimal set of attempt `attempt` (core/cnn_softam.h:1021-1039): candidate cells k = 0, 1, 2, ... until four distinct ones.
// false: more than 32 candidates were needed (degenerate tiny maps).  The first four candidates are drawn as straight-line code (four
// independent 64-bit mixes whose quarter-rate multiplies overlap); the loop with the reference's redraw-on-duplicate semantics only
// runs when two of them coincide (probability ~6/P).
DM_INLINE bool draw_set(const FrameDev& F, uint64_t key, uint32_t attempt, int32_t set4[4]) {
    const uint32_t W = (uint32_t)F.W, H = (uint32_t)F.H;
#pragma unroll
    for (int j = 0; j < 4; j++) set4[j] = dm::draw_cell(key, attempt, (uint32_t)j, W, H);
    const bool distinct = set4[0] != set4[1] && set4[0] != set4[2] && set4[0] != set4[3] && set4[1] != set4[2] && set4[1] != set4[3] && set4[2] != set4[3];
    if (distinct) return true;
    uint32_t k = 0;
    int cnt = 0;
    set4[0] = set4[1] = set4[2] = set4[3] = 0;
    while (cnt < 4) {
        if (k >= 32) return false;
        const int idx = dm::draw_cell(key, attempt, k++, W, H);
        const bool dup = (cnt > 0 && set4[0] == idx) || (cnt > 1 && set4[1] == idx) || (cnt > 2 && set4[2] == idx);
        if (dup) continue;
        if (cnt == 0) set4[0] = idx; else if (cnt == 1) set4[1] = idx; else if (cnt == 2) set4[2] = idx; else set4[3] = idx;
        cnt++;
    }
    return true;
}

// One wave per hypothesis.  Lane l evaluates quartic root (l & 3) of attempt base + (l >> 2): 16 attempts
// per round, the four candidate poses of an attempt side by side (their Jacobi eigen-solves, the long pole
// of P3P, run in parallel instead of in sequence).
// WPB waves (hypotheses) per workgroup.  A K1 wave holds 288 registers (occupancy 1).  Round 1 ran K1 underneath the bandwidth-bound K2
// of another frame and chose 4 waves per workgroup (one CU in four, all four SIMDs: 90.3 vs 91.3 us per step with 1, 94.6 with 8 = 2 per
// SIMD and 100 B of scratch, 125 with 16).  Round 2's default step does not overlap (with K2 at 0.70 of HBM peak the overlap no longer
// pays, profiles/r02_bench_modes.txt) and K1 alone is fastest with ONE wave per workgroup -- every CU gets work first:
// N = 2048: 48.6 us (1) / 54.8 (4) / 49.4 (8) / 50.0 (register budget for 2 waves per SIMD, 140 B scratch); N = 4096: 66.8 / 83.7 / 76.4 /
// 63.1 (profiles/r02_k1_variants.txt).  Raised wave priority is worth ~1 us under K2.
// HPW hypotheses per wave: each hypothesis owns 64/HPW lanes = 16/HPW attempts per round (4 root lanes each).  Attempts are
// consumed in index order and the lowest accepted index wins, so the result does not depend on HPW (the GPU tests pass with
// DSAC_K1_HPW=2 and 4).  The idea was that the instruction stream of a round costs the same whether 4 or 64 lanes are live;
// measured (scripts/k1_bench.py, 640x480 synthetic frame): N=2048 takes 65 us with HPW=1, 91 us with 2, 183 us with 4 -- a
// hypothesis needs ~20 attempts here (4 noisy inliers rarely re-project the 4th point within 10 px), so the 16 attempt lanes
// of HPW=1 are busy and fewer lanes per hypothesis only serialise.  HPW=1 stays the default; the knob is for easy frames.
// HORN: align the P3P triangle by OpenCV's own iteration -- Horn's quaternion method through Jacobi sweeps of the 4x4 matrix -- instead of the
// closed form of the same least-squares optimum (dmath.h align3_lsq, the default since round 5; rounds 1-4: an orthonormal triad): 43 -> 230 us at
// 4096 hypotheses, kept as the literal restatement for comparisons.
// MINW: minimum waves per SIMD the register allocation must leave room for (1: 288 registers, no scratch; 2: 256 registers + 140 B of
// scratch per lane, two waves share a SIMD and hide each other's fp64 dependency chains)
// RL: lanes per attempt.  4 = one lane per quartic root (16 attempts per round and hypothesis); 1 = one lane per ATTEMPT, its roots in
// sequence (64 attempts per round): a round is longer (set-up + up to 4 root evaluations instead of one) but four times as wide, so a
// hypothesis rarely needs a second one -- the launch time is the slowest hypothesis' number of rounds.
template <int WPB, int HPW, bool HORN = false, int MINW = 1, int RL = 4>
__global__ __launch_bounds__(64 * WPB, MINW) void k_sample(int N, uint64_t seed, FrameDev F, int thr_int, int max_tries, double* __restrict__ poses,
                                                     int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok, float* __restrict__ staged, int prio,
                                                     int Nf) {
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    constexpr int LPH = 64 / HPW;        // lanes per hypothesis
    constexpr int APR = LPH / RL;        // attempts per round and hypothesis
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPH, sl = lane % LPH;
    const int h = (blockIdx.x * WPB + (threadIdx.x >> 6)) * HPW + sub;
    const unsigned long long submask = (HPW == 1) ? ~0ull : (((1ull << LPH) - 1ull) << (sub * LPH));
    bool done = h >= N;                  // uniform over the lanes of a hypothesis
    const int hc = done ? 0 : h;
    const int root = lane & (RL - 1);
    const int frame = hc / Nf;           // Nf == N for a single frame
    F.xyz += (long long)frame * F.xyz_stride;
    if (F.uv) F.uv += (long long)frame * F.uv_stride;
    const uint64_t key = dm::hyp_key(seed + (uint64_t)frame * (uint64_t)F.seed_stride, (uint32_t)(hc - frame * Nf));
    const dm::Cam K = make_cam(F);
    for (int base = 0; base < max_tries; base += APR) {
        if (__ballot(!done) == 0ull) return;
        const uint32_t attempt = (uint32_t)(base + sl / RL);
        int32_t set4[4];
        bool live = !done && (int)attempt < max_tries;
        if (live) live = draw_set(F, key, attempt, set4);
        float X[4][3], uv[4][2];
        double Rc[9], Tc[3], reproj = 0;
        bool cand = false;
        if (live) {
            load_set(F, set4, X, uv);
            dm::P3PSetup S;
            if (RL == 4) {
                if (dm::p3p_setup(X, uv, K, S) && root < S.n) {
                    const double x = (root == 0) ? S.roots[0] : (root == 1) ? S.roots[1] : (root == 2) ? S.roots[2] : S.roots[3];
                    cand = dm::p3p_eval_root<HORN>(S, X, uv, K, x, Rc, Tc, reproj);
                }
            } else if (dm::p3p_setup(X, uv, K, S)) {
                // the roots in sequence, the one whose pose re-projects the 4th point best wins, the first on ties (dm::p3p)
                for (int i = 0; i < 4; i++) {
                    if (i >= S.n) continue;
                    double Rt[9], Tt[3], rp;
                    const double x = (i == 0) ? S.roots[0] : (i == 1) ? S.roots[1] : (i == 2) ? S.roots[2] : S.roots[3];
                    if (!dm::p3p_eval_root<HORN>(S, X, uv, K, x, Rt, Tt, rp)) continue;
                    if (!cand || reproj > rp) {
                        cand = true;
                        reproj = rp;
#pragma unroll
                        for (int k = 0; k < 9; k++) Rc[k] = Rt[k];
                        Tc[0] = Tt[0]; Tc[1] = Tt[1]; Tc[2] = Tt[2];
                    }
                }
            }
        }
        const int win = (RL == 4) ? dm::best_root_of_quad(cand, reproj) : (cand ? 0 : -1);
        bool good = false;
        double cv6[6] = {0, 0, 0, 0, 0, 0};
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        // Early rejection: `reproj` is the squared re-projection error (px^2, fp64) of the 4th point under (Rc, Tc); the check
        // below re-derives the rotation through the Rodrigues vector and rounds the projections to float, which moves that
        // error by < 1e-3 px.  An attempt whose 4th point misses the threshold by more than 0.01 px can therefore not pass, and
        // a round in which no lane can pass (the common case: ~20 attempts per accepted set) skips the Rodrigues round trip and
        // the four projections altogether.
        const double thr_hi = (double)thr_int + 0.01;
        if (live && win == root && !(reproj > thr_hi * thr_hi)) {
            dm::rodrigues_m2v(Rc, cv6);
            cv6[3] = Tc[0]; cv6[4] = Tc[1]; cv6[5] = Tc[2];
            // 4-point re-projection check (core/cnn_softam.h:1046-1059), through Rodrigues(rvec) like projectPoints
            dm::rodrigues_v2m<false>(cv6, R, nullptr);
            good = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float u, v;
                dm::project_f(R, cv6 + 3, K, X[j][0], X[j][1], X[j][2], u, v);
                const float dx = uv[j][0] - u, dy = uv[j][1] - v;
                good = good && (sqrt((double)dx * dx + (double)dy * dy) < (double)thr_int);
            }
        }
        const unsigned long long m = __ballot(good) & submask;
        if (m != 0ull) {
            const int w = __ffsll((long long)m) - 1;  // lowest lane of this hypothesis = lowest attempt index
            if (lane == w) {
#pragma unroll
                for (int k = 0; k < 6; k++) poses[(size_t)h * 6 + k] = cv6[k];
#pragma unroll
                for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = set4[k];
                ok[h] = 1;
                if (staged) write_staged_R(F, R, cv6, staged + (size_t)h * POSE_STRIDE);
            }
            done = true;
        }
    }
    // no accepted attempt: zero pose, ok = 0; sets_out reports the last attempt's set
    if (!done && sl == 0) {
        int32_t set4[4];
        draw_set(F, key, (uint32_t)(max_tries - 1), set4);
#pragma unroll
        for (int kk = 0; kk < 6; kk++) poses[(size_t)h * 6 + kk] = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) sets_out[(size_t)h * 4 + kk] = set4[kk];
        ok[h] = 0;
        if (staged) { const double z6[6] = {0, 0, 0, 0, 0, 0}; write_staged(F, z6, staged + (size_t)h * POSE_STRIDE); }
    }
}

// --------------------------------------------------------------------------------------------------
// K1 for few hypotheses (one frame): WPH waves per hypothesis from the first round on, one lane per attempt, so a round is 64 * WPH
// attempts wide.  With one wave (64 attempts, failure probability ~4 % on the synthetic frames) some of 256 hypotheses need a second
// round in practically every launch, and the launch lasts as long as its slowest hypothesis; with four waves (256 attempts) a second
// round is a 1e-6 event, and 256 x 4 waves are exactly the chip's 1024 SIMDs -- measured (profiles/r02_k1_wide.txt), two waves are the
// better trade: N = 256 15.1 -> 11.4 us (four: 12.3), N = 512 16.3 -> 11.8 us; the launcher uses two up to 512 hypotheses.  Same result: wave w takes attempts base + 64 w + lane,
// the lowest accepted attempt of the round wins (LDS exchange, two barriers per round), rounds are consumed in increasing order.
// --------------------------------------------------------------------------------------------------
template <int WPH>
__global__ __launch_bounds__(64 * WPH) void k_sample_wide(int N, uint64_t seed, FrameDev F, int thr_int, int max_tries, double* __restrict__ poses,
                                                         int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok, float* __restrict__ staged, int prio,
                                                         int Nf) {
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    __shared__ int s_win[WPH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x;  // grid = N
    const int frame = h / Nf;
    F.xyz += (long long)frame * F.xyz_stride;
    if (F.uv) F.uv += (long long)frame * F.uv_stride;
    const uint64_t key = dm::hyp_key(seed + (uint64_t)frame * (uint64_t)F.seed_stride, (uint32_t)(h - frame * Nf));
    const dm::Cam K = make_cam(F);
    const double thr_hi = (double)thr_int + 0.01;
    bool done = false;
    for (long long base = 0; base < (long long)max_tries && !done; base += 64 * WPH) {
        const long long att = base + 64 * wave + lane;
        const uint32_t attempt = (uint32_t)att;
        int32_t set4[4];
        bool live = att < (long long)max_tries;
        if (live) live = draw_set(F, key, attempt, set4);
        float X[4][3], uv[4][2];
        double Rc[9], Tc[3], reproj = 0;
        bool cand = false;
        if (live) {
            load_set(F, set4, X, uv);
            dm::P3PSetup S;
            if (dm::p3p_setup(X, uv, K, S)) {
                for (int i = 0; i < 4; i++) {  // the roots in sequence, best 4th-point re-projection wins, the first on ties (dm::p3p)
                    if (i >= S.n) continue;
                    double Rt[9], Tt[3], rp;
                    const double x = (i == 0) ? S.roots[0] : (i == 1) ? S.roots[1] : (i == 2) ? S.roots[2] : S.roots[3];
                    if (!dm::p3p_eval_root<false>(S, X, uv, K, x, Rt, Tt, rp)) continue;
                    if (!cand || reproj > rp) {
                        cand = true;
                        reproj = rp;
#pragma unroll
                        for (int k = 0; k < 9; k++) Rc[k] = Rt[k];
                        Tc[0] = Tt[0]; Tc[1] = Tt[1]; Tc[2] = Tt[2];
                    }
                }
            }
        }
        bool good = false;
        double cv6[6] = {0, 0, 0, 0, 0, 0};
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (live && cand && !(reproj > thr_hi * thr_hi)) {  // early rejection as in k_sample
            dm::rodrigues_m2v(Rc, cv6);
            cv6[3] = Tc[0]; cv6[4] = Tc[1]; cv6[5] = Tc[2];
            dm::rodrigues_v2m<false>(cv6, R, nullptr);
            good = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float uu, vv;
                dm::project_f(R, cv6 + 3, K, X[j][0], X[j][1], X[j][2], uu, vv);
                const float dx = uv[j][0] - uu, dy = uv[j][1] - vv;
                good = good && (sqrt((double)dx * dx + (double)dy * dy) < (double)thr_int);
            }
        }
        const unsigned long long m = __ballot(good);
        const int wl = m ? (__ffsll((long long)m) - 1) : -1;
        if (lane == 0) s_win[wave] = m ? 64 * wave + wl : 0x7fffffff;
        __syncthreads();
        int best = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < WPH; w++) best = min(best, s_win[w]);
        if (best != 0x7fffffff) {
            done = true;
            if (best == 64 * wave + lane) {  // exactly one lane of the workgroup
#pragma unroll
                for (int k = 0; k < 6; k++) poses[(size_t)h * 6 + k] = cv6[k];
#pragma unroll
                for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = set4[k];
                ok[h] = 1;
                if (staged) write_staged_R(F, R, cv6, staged + (size_t)h * POSE_STRIDE);
            }
        }
        __syncthreads();  // s_win is rewritten in the next round
    }
    // no accepted attempt: zero pose, ok = 0; sets_out reports the last attempt's set
    if (!done && threadIdx.x == 0) {
        int32_t set4[4];
        draw_set(F, key, (uint32_t)(max_tries - 1), set4);
#pragma unroll
        for (int kk = 0; kk < 6; kk++) poses[(size_t)h * 6 + kk] = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) sets_out[(size_t)h * 4 + kk] = set4[kk];
        ok[h] = 0;
        if (staged) { const double z6[6] = {0, 0, 0, 0, 0, 0}; write_staged(F, z6, staged + (size_t)h * POSE_STRIDE); }
    }
}

// --------------------------------------------------------------------------------------------------
// K1 with work sharing inside a workgroup (round 2).  The launch time of k_sample is the time of its SLOWEST hypothesis: an attempt is
// accepted with p ~ 0.05 on the synthetic frames, a round of 16 fails with 0.44, so the slowest of 2048 hypotheses needs ~9 rounds of
// ~5.5 us while the average needs 1.8.  Here SW waves form a workgroup over SW consecutive hypotheses; a wave whose hypothesis is
// accepted (or exhausted) joins one of the workgroup's unfinished hypotheses in the next round and evaluates the NEXT 16 attempts of
// that hypothesis.  A round therefore consumes a contiguous block of 16 x (waves on it) attempt indices per hypothesis, the lowest
// accepted index of the block wins, and blocks are consumed in increasing order -- the same "first accepted attempt in index order"
// the one-wave form and the reference's sequential loop produce, independent of SW and of scheduling (the assignment of helpers is a
// function of the done-state at the start of the round, exchanged through LDS with two barriers per round).
// Measured (profiles/r02_k1_share.txt, 640x480 synthetic frame): N = 256 34.8 -> 29.2 us, N = 1024 36.3 -> 30.3 us; but N = 2048 48.4 -> 51.6
// and N = 4096 66.9 -> 83.8 us -- once there are more waves than SIMDs the launch is throughput-bound and 4-wave workgroups at one wave
// per SIMD place worse than single waves (and a register budget for two waves per SIMD costs scratch: 62.5 us at N = 2048).  The launcher
// therefore shares only up to 1024 hypotheses (one frame, or a small batch).
// --------------------------------------------------------------------------------------------------
template <int SW>
__global__ __launch_bounds__(64 * SW) void k_sample_shared(int N, uint64_t seed, FrameDev F, int thr_int, int max_tries, double* __restrict__ poses,
                                                           int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok, float* __restrict__ staged, int prio,
                                                           int Nf) {
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    __shared__ int s_fin[SW];   // hypothesis w of the workgroup is finished: accepted, exhausted or beyond N
    __shared__ int s_acc[SW];   // ... and was accepted
    __shared__ int s_base[SW];  // next attempt index of hypothesis w
    __shared__ int s_win[SW];   // lowest accepted attempt index found by wave w in this round (INT_MAX: none)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h_first = blockIdx.x * SW;
    if (threadIdx.x < SW) {
        s_fin[threadIdx.x] = (h_first + (int)threadIdx.x >= N) ? 1 : 0;
        s_acc[threadIdx.x] = 0;
        s_base[threadIdx.x] = 0;
    }
    __syncthreads();
    const int root = lane & 3;
    const dm::Cam K = make_cam(F);
    const float* xyz0 = F.xyz;
    const float* uv0 = F.uv;
    const double thr_hi = (double)thr_int + 0.01;
    for (;;) {
        // snapshot of the workgroup's state (uniform)
        int fin[SW], base[SW];
        int u = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) { fin[w] = s_fin[w]; base[w] = s_base[w]; u += fin[w] ? 0 : 1; }
        if (u == 0) break;
        // target of every wave: its own hypothesis while unfinished; the k-th finished wave helps the (k mod u)-th unfinished hypothesis
        int tgt[SW], slot[SW];
        {
            int unf[SW], nu = 0, k = 0;
#pragma unroll
            for (int w = 0; w < SW; w++) if (!fin[w]) unf[nu++] = w;
#pragma unroll
            for (int w = 0; w < SW; w++) {
                if (!fin[w]) { tgt[w] = w; slot[w] = 0; }
                else { tgt[w] = unf[k % nu]; slot[w] = 1 + k / nu; k++; }
            }
        }
        int t = 0, sl = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) if (w == wave) { t = tgt[w]; sl = slot[w]; }
        int bt = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) if (w == t) bt = base[w];
        const int h = h_first + t;
        const int frame = h / Nf;
        F.xyz = xyz0 + (long long)frame * F.xyz_stride;
        F.uv = uv0 ? uv0 + (long long)frame * F.uv_stride : nullptr;
        const uint64_t key = dm::hyp_key(seed + (uint64_t)frame * (uint64_t)F.seed_stride, (uint32_t)(h - frame * Nf));
        const long long att = (long long)bt + 16ll * sl + (lane >> 2);
        const uint32_t attempt = (uint32_t)att;
        int32_t set4[4];
        bool live = att < (long long)max_tries;
        if (live) live = draw_set(F, key, attempt, set4);
        float X[4][3], uv[4][2];
        double Rc[9], Tc[3], reproj = 0;
        bool cand = false;
        if (live) {
            load_set(F, set4, X, uv);
            dm::P3PSetup S;
            if (dm::p3p_setup(X, uv, K, S) && root < S.n) {
                const double x = (root == 0) ? S.roots[0] : (root == 1) ? S.roots[1] : (root == 2) ? S.roots[2] : S.roots[3];
                cand = dm::p3p_eval_root<false>(S, X, uv, K, x, Rc, Tc, reproj);
            }
        }
        const int win = dm::best_root_of_quad(cand, reproj);
        bool good = false;
        double cv6[6] = {0, 0, 0, 0, 0, 0};
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (live && win == root && !(reproj > thr_hi * thr_hi)) {  // early rejection as in k_sample
            dm::rodrigues_m2v(Rc, cv6);
            cv6[3] = Tc[0]; cv6[4] = Tc[1]; cv6[5] = Tc[2];
            dm::rodrigues_v2m<false>(cv6, R, nullptr);
            good = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float uu, vv;
                dm::project_f(R, cv6 + 3, K, X[j][0], X[j][1], X[j][2], uu, vv);
                const float dx = uv[j][0] - uu, dy = uv[j][1] - vv;
                good = good && (sqrt((double)dx * dx + (double)dy * dy) < (double)thr_int);
            }
        }
        const unsigned long long m = __ballot(good);
        const int wl = m ? (__ffsll((long long)m) - 1) : -1;  // lowest accepted lane of this wave = its lowest accepted attempt
        const int mine = m ? (bt + 16 * sl + (wl >> 2)) : 0x7fffffff;
        if (lane == 0) s_win[wave] = mine;
        __syncthreads();
        int best_t = 0x7fffffff, n_t = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) if (tgt[w] == t) { best_t = min(best_t, s_win[w]); n_t++; }
        if (mine == best_t && mine != 0x7fffffff && lane == wl) {  // attempt indices are unique per (hypothesis, wave): exactly one writer
#pragma unroll
            for (int k = 0; k < 6; k++) poses[(size_t)h * 6 + k] = cv6[k];
#pragma unroll
            for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = set4[k];
            ok[h] = 1;
            if (staged) write_staged_R(F, R, cv6, staged + (size_t)h * POSE_STRIDE);
        }
        if (t == wave && lane == 0) {  // the owner advances its hypothesis
            const int nb = bt + 16 * n_t;
            s_base[wave] = nb;
            if (best_t != 0x7fffffff) { s_fin[wave] = 1; s_acc[wave] = 1; }
            else if (nb >= max_tries) s_fin[wave] = 1;
        }
        __syncthreads();
    }
    // no accepted attempt: zero pose, ok = 0; sets_out reports the last attempt's set
    const int h = h_first + wave;
    if (h < N && !s_acc[wave] && lane == 0) {
        const int frame = h / Nf;
        F.xyz = xyz0 + (long long)frame * F.xyz_stride;
        F.uv = uv0 ? uv0 + (long long)frame * F.uv_stride : nullptr;
        const uint64_t key = dm::hyp_key(seed + (uint64_t)frame * (uint64_t)F.seed_stride, (uint32_t)(h - frame * Nf));
        int32_t set4[4];
        draw_set(F, key, (uint32_t)(max_tries - 1), set4);
#pragma unroll
        for (int kk = 0; kk < 6; kk++) poses[(size_t)h * 6 + kk] = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) sets_out[(size_t)h * 4 + kk] = set4[kk];
        ok[h] = 0;
        if (staged) { const double z6[6] = {0, 0, 0, 0, 0, 0}; write_staged(F, z6, staged + (size_t)h * POSE_STRIDE); }
    }
}

// Given sets: one lane per hypothesis.
template <bool HORN>
__global__ __launch_bounds__(64) void k_eval_sets(int N, const int32_t* __restrict__ sets_in, FrameDev F, int thr_int,
                                                  double* __restrict__ poses, int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok,
                                                  float* __restrict__ staged) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    int32_t set4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) set4[k] = sets_in[(size_t)h * 4 + k];
    double cv6[6] = {0, 0, 0, 0, 0, 0};
    const bool good = solve_and_check<HORN>(F, set4, thr_int, cv6);
#pragma unroll
    for (int k = 0; k < 6; k++) poses[(size_t)h * 6 + k] = good ? cv6[k] : 0.0;
    if (sets_out != sets_in) {
#pragma unroll
        for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = set4[k];
    }
    ok[h] = good ? 1 : 0;
    if (staged) {
        if (!good) { cv6[0] = cv6[1] = cv6[2] = cv6[3] = cv6[4] = cv6[5] = 0.0; }
        write_staged(F, cv6, staged + (size_t)h * POSE_STRIDE);
    }
}

hipError_t sample(hipStream_t st, int N, uint64_t seed, const int32_t* sets_in, const FrameDev& F, int thr_int, int max_tries, double* poses,
                  int32_t* sets_out, uint8_t* ok, float* staged, int Nf, const K1Opts& o) {
    if (N <= 0) return hipSuccess;
    if (sets_in && Nf > 0 && F.frames > 1) return hipErrorInvalidValue;  // given sets are evaluated on one frame only
    if (sets_in) {
        if (o.horn) hipLaunchKernelGGL(k_eval_sets<true>, dim3((N + 63) / 64), dim3(64), 0, st, N, sets_in, F, thr_int, poses, sets_out, ok, staged);
        else hipLaunchKernelGGL(k_eval_sets<false>, dim3((N + 63) / 64), dim3(64), 0, st, N, sets_in, F, thr_int, poses, sets_out, ok, staged);
    } else {
        const int wpb = o.wpb, prio = o.prio;
        const int H2 = o.hpw > 0 ? o.hpw : 1;  // see the measurement in the kernel's comment
#define DSAC_K1(W, G, HN) hipLaunchKernelGGL((k_sample<W, G, HN>), dim3((N + W * G - 1) / (W * G)), dim3(64 * W), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, Nf > 0 ? Nf : N)
        const int NfK = Nf > 0 ? Nf : N;
        if (o.horn) DSAC_K1(1, 1, true);
        else if (o.rl != 1 && o.share >= 2 && H2 == 1 && (N <= 1024 || o.share_always) && (F.frames <= 1 || NfK % o.share == 0)) {  // a workgroup's hypotheses must share a frame
            if (o.share >= 8) hipLaunchKernelGGL((k_sample_shared<8>), dim3((N + 7) / 8), dim3(512), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
            else hipLaunchKernelGGL((k_sample_shared<4>), dim3((N + 3) / 4), dim3(256), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        }
        else if (o.minw >= 2 && H2 == 1 && o.rl == 1 && wpb < 4)  // one lane per attempt at a register budget for two waves per SIMD: 292 B of scratch,
                                                                  // 61 against 43 us at N = 4096 (profiles/r04_k1_minw.txt) -- a measurement knob, not a default
            hipLaunchKernelGGL((k_sample<1, 1, false, 2, 1>), dim3(N), dim3(64), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else if (o.minw >= 2 && H2 == 1) {
            if (wpb >= 4) hipLaunchKernelGGL((k_sample<4, 1, false, 2>), dim3((N + 3) / 4), dim3(256), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, Nf > 0 ? Nf : N);
            else hipLaunchKernelGGL((k_sample<1, 1, false, 2>), dim3(N), dim3(64), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, Nf > 0 ? Nf : N);
        }
        else if (o.rl == 1 && H2 == 2) hipLaunchKernelGGL((k_sample<1, 2, false, 1, 1>), dim3((N + 1) / 2), dim3(64), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else if (o.rl == 1 && H2 == 4) hipLaunchKernelGGL((k_sample<1, 4, false, 1, 1>), dim3((N + 3) / 4), dim3(64), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else if (H2 >= 4) { if (wpb >= 4) DSAC_K1(4, 4, false); else DSAC_K1(1, 4, false); }
        else if (H2 == 2) { if (wpb >= 8) DSAC_K1(8, 2, false); else if (wpb >= 4) DSAC_K1(4, 2, false); else DSAC_K1(1, 2, false); }
        else if (o.rl == 1 && H2 == 1 && o.minw < 2 && wpb < 4 && o.wide == 4 && N <= 256)
            hipLaunchKernelGGL((k_sample_wide<4>), dim3(N), dim3(256), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else if (o.rl == 1 && H2 == 1 && o.minw < 2 && wpb < 4 && o.wide != 0 && N <= 512 && (o.wide < 0 || o.wide == 2))
            // 56 KiB of unused dynamic LDS per two-wave workgroup: at most two workgroups per CU = one wave per SIMD.  Since round 5 the kernel needs 232
            // registers (no hoisted libm constants), two of its waves fit a SIMD -- and the dispatcher then packs up to 512 hypotheses x 2 waves onto half
            // the chip: 16.1 against 12.1 us at N = 512 (profiles/r05_k1_licm_ab.txt).  With few hypotheses every wave wants a SIMD of its own
            hipLaunchKernelGGL((k_sample_wide<2>), dim3(N), dim3(128), 56 * 1024, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else if (o.rl == 1 && H2 == 1 && o.minw < 2 && wpb < 4) hipLaunchKernelGGL((k_sample<1, 1, false, 1, 1>), dim3(N), dim3(64), 0, st, N, seed, F, thr_int, max_tries, poses, sets_out, ok, staged, prio, NfK);
        else { if (wpb >= 8) DSAC_K1(8, 1, false); else if (wpb >= 4) DSAC_K1(4, 1, false); else DSAC_K1(1, 1, false); }
#undef DSAC_K1
    }
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K1 in the REFERENCE'S OWN random stream (round 6; refstream.h has the stream's arithmetic).  The reference serves the hypotheses of OpenMP thread t one
// after the other from std::mt19937(seed + t): which attempt draws which cells is a function of the stream alone, only WHICH attempt goes to which hypothesis
// depends on P3P.  So: (1) one workgroup per stream generates a window of raw outputs (the MT19937 twist in three data-parallel phases) and parses it into
// attempts -- a wave at a time, 64 attempts speculatively at 8 outputs each, the valid prefix ends at the first attempt that needed more (a duplicate cell,
// a rejected draw); (2) every attempt of every stream is evaluated in parallel by K1's own solve_and_check, one lane per attempt; (3) one workgroup per
// stream prefix-counts the accepted flags: the j-th accepted attempt is the stream's j-th hypothesis, exactly what the sequential loop would have kept;
// the generator advances to the output behind the last attempt used.  A window that does not serve every hypothesis is followed by another (host loop).
// --------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;

__global__ __launch_bounds__(64) void k_refstream_init(RefStreamState* st, uint32_t seed, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    rs::mt_seed(st[t].mt, seed + (uint32_t)t);
    st[t].idx = rs::MT_N;  // std::mt19937 after seed(): the first output twists
}

// the next block in place, by the whole workgroup (mt in LDS).  x[k + n] needs x[k], x[k + 1], x[k + m]: k < 227 read only old words, 227 <= k < 454 the new
// words of the first phase, the rest those of the second
DM_INLINE void rs_twist_lds(uint32_t* mt, int tid) {
    constexpr int D = rs::MT_N - rs::MT_M;  // 227
    for (int ph = 0; ph < 3; ph++) {
        const int k = ph * D + tid;
        const bool act = tid < D && k < rs::MT_N;
        uint32_t v = 0;
        if (act) v = rs::mt_twist_word(mt[k], mt[k + 1 == rs::MT_N ? 0 : k + 1], ph == 0 ? mt[k + rs::MT_M] : mt[k - D]);
        __syncthreads();
        if (act) mt[k] = v;
        __syncthreads();
    }
}

// advance a generator by n outputs (std::mt19937::discard)
DM_INLINE void rs_advance_lds(uint32_t* mt, uint32_t& idx, unsigned long long n, int tid) {
    const unsigned long long total = (unsigned long long)idx + n;
    if (total <= (unsigned long long)rs::MT_N) { idx = (uint32_t)total; return; }
    const unsigned long long twists = (total - 1) / rs::MT_N;
    for (unsigned long long i = 0; i < twists; i++) rs_twist_lds(mt, tid);
    idx = (uint32_t)(total - twists * rs::MT_N);
}

__global__ __launch_bounds__(RS_THREADS) void k_refstream_discard(RefStreamState* st, int t, unsigned long long n) {
    __shared__ uint32_t mt[rs::MT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < rs::MT_N; i += RS_THREADS) mt[i] = st[t].mt[i];
    uint32_t idx = st[t].idx;
    __syncthreads();
    rs_advance_lds(mt, idx, n, tid);
    for (int i = tid; i < rs::MT_N; i += RS_THREADS) st[t].mt[i] = mt[i];
    if (tid == 0) st[t].idx = idx;
}

// (1) window of stream t = blockIdx.x: raw[t][0 .. D) and its attempts.  need[t] == 0: nothing to do (parsed = 0).
__global__ __launch_bounds__(RS_THREADS) void k_refstream_parse(const RefStreamState* __restrict__ st, int W, int H, int mode, int A, int D, uint32_t* __restrict__ raw_all,
                                                                int32_t* __restrict__ sets_all, uint32_t* __restrict__ offs_all, int32_t* __restrict__ parsed,
                                                                const int32_t* __restrict__ need) {
    const int t = blockIdx.x, tid = threadIdx.x;
    if (need[t] <= 0) { if (tid == 0) parsed[t] = 0; return; }
    __shared__ uint32_t mt[rs::MT_N];
    uint32_t* raw = raw_all + (size_t)t * D;
    for (int i = tid; i < rs::MT_N; i += RS_THREADS) mt[i] = st[t].mt[i];
    const int idx0 = (int)st[t].idx;
    __syncthreads();
    int o = 0;
    for (int i = idx0 + tid; i < rs::MT_N && o + (i - idx0) < D; i += RS_THREADS) raw[o + (i - idx0)] = rs::mt_temper(mt[i]);
    o += rs::MT_N - idx0;
    while (o < D) {
        rs_twist_lds(mt, tid);
        for (int i = tid; i < rs::MT_N && o + i < D; i += RS_THREADS) raw[o + i] = rs::mt_temper(mt[i]);
        o += rs::MT_N;
    }
    __threadfence_block();
    __syncthreads();
    if (tid >= 64) return;
    int32_t* sets = sets_all + (size_t)t * A * 4;
    uint32_t* offs = offs_all + (size_t)t * (A + 1);
    const int lane = tid;
    long long cur = 0;
    int a = 0;
    while (a < A) {
        const long long start = cur + 8 * lane;
        int32_t set4[4];
        const int len = rs::parse_attempt([&](long long i) { return raw[i]; }, start, (long long)D, (uint32_t)W, (uint32_t)H, mode, set4);
        const unsigned long long odd = __ballot(len != 8);
        const int k0 = odd ? __ffsll((long long)odd) - 1 : 64;
        const int len_k0 = __shfl(len, k0 & 63, 64);
        int nv = odd ? (len_k0 > 0 ? k0 + 1 : k0) : 64;  // the attempts of lanes 0 .. k0 start where the speculation put them; a starved / degenerate one is not an attempt
        nv = min(nv, A - a);
        if (lane < nv) {
#pragma unroll
            for (int k = 0; k < 4; k++) sets[(size_t)(a + lane) * 4 + k] = set4[k];
            offs[a + lane] = (uint32_t)start;
        }
        if (nv == 0) break;
        const long long st_last = cur + 8 * (nv - 1);
        const int len_last = __shfl(len, nv - 1, 64);
        cur = st_last + len_last;
        a += nv;
        if (odd && len_k0 <= 0) break;
    }
    if (lane == 0) { offs[a] = (uint32_t)cur; parsed[t] = a; }
}

// (2) one lane per attempt
__global__ __launch_bounds__(64) void k_refstream_eval(int A, const int32_t* __restrict__ parsed, const int32_t* __restrict__ sets_all, FrameDev F, int thr_int,
                                                       double* __restrict__ poses_tmp, uint8_t* __restrict__ ok_tmp) {
    const int t = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= parsed[t]) return;
    const size_t i = (size_t)t * A + a;
    int32_t set4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) set4[k] = sets_all[i * 4 + k];
    double cv6[6] = {0, 0, 0, 0, 0, 0};
    const bool good = solve_and_check<false>(F, set4, thr_int, cv6);
#pragma unroll
    for (int k = 0; k < 6; k++) poses_tmp[i * 6 + k] = good ? cv6[k] : 0.0;
    ok_tmp[i] = good ? 1 : 0;
}

// (3) stream t = blockIdx.x: its next need[t] hypotheses are its first need[t] accepted attempts of the window
__global__ __launch_bounds__(RS_THREADS) void k_refstream_select(RefStreamState* __restrict__ st, int A, const int32_t* __restrict__ parsed, const int32_t* __restrict__ sets_all,
                                                                 const uint32_t* __restrict__ offs_all, const double* __restrict__ poses_tmp, const uint8_t* __restrict__ ok_tmp,
                                                                 const int32_t* __restrict__ first, int32_t* __restrict__ served, int32_t* __restrict__ need,
                                                                 unsigned long long* __restrict__ consumed, long long* __restrict__ attempts, FrameDev F,
                                                                 double* __restrict__ poses, int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok, float* __restrict__ staged) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int n = parsed[t], want = need[t];
    if (want <= 0 || n <= 0) return;
    __shared__ int s_cnt[RS_THREADS];
    __shared__ int s_last;
    __shared__ uint32_t mt[rs::MT_N];
    const size_t base = (size_t)t * A;
    const int per = (n + RS_THREADS - 1) / RS_THREADS, a0 = tid * per, a1 = min(n, a0 + per);
    int cnt = 0;
    for (int a = a0; a < a1; a++) cnt += ok_tmp[base + a];
    s_cnt[tid] = cnt;
    if (tid == 0) s_last = -1;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < RS_THREADS; i++) { const int v = s_cnt[i]; if (i < tid) before += v; total += v; }
    const int row0 = first[t] + served[t];
    int r = before;
    for (int a = a0; a < a1 && r < want; a++) {
        if (!ok_tmp[base + a]) continue;
        const int h = row0 + r;
        double cv6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { cv6[k] = poses_tmp[(base + a) * 6 + k]; poses[(size_t)h * 6 + k] = cv6[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = sets_all[(base + a) * 4 + k];
        ok[h] = 1;
        if (staged) write_staged(F, cv6, staged + (size_t)h * POSE_STRIDE);
        if (r == want - 1) s_last = a;
        r++;
    }
    __syncthreads();
    const int got = min(total, want);
    const int used = total >= want ? s_last + 1 : n;  // every attempt of the window is spent when it did not serve the stream
    const unsigned long long took = offs_all[(size_t)t * (A + 1) + used];
    for (int i = tid; i < rs::MT_N; i += RS_THREADS) mt[i] = st[t].mt[i];
    uint32_t idx = st[t].idx;
    __syncthreads();
    rs_advance_lds(mt, idx, took, tid);
    for (int i = tid; i < rs::MT_N; i += RS_THREADS) st[t].mt[i] = mt[i];
    if (tid == 0) {
        st[t].idx = idx;
        served[t] += got;
        need[t] = want - got;
        consumed[t] += took;
        attempts[t] += used;
    }
}

// hypotheses a stream could not serve within the attempt budget: zero pose, ok = 0 (as K1 reports a hypothesis that ran out of tries)
__global__ __launch_bounds__(64) void k_refstream_unserved(int T, const int32_t* __restrict__ first, const int32_t* __restrict__ served, const int32_t* __restrict__ need, FrameDev F,
                                                           double* __restrict__ poses, int32_t* __restrict__ sets_out, uint8_t* __restrict__ ok, float* __restrict__ staged) {
    const int t = blockIdx.x;
    for (int j = threadIdx.x; j < need[t]; j += blockDim.x) {
        const int h = first[t] + served[t] + j;
#pragma unroll
        for (int k = 0; k < 6; k++) poses[(size_t)h * 6 + k] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) sets_out[(size_t)h * 4 + k] = 0;
        ok[h] = 0;
        if (staged) { const double z6[6] = {0, 0, 0, 0, 0, 0}; write_staged(F, z6, staged + (size_t)h * POSE_STRIDE); }
    }
}

hipError_t refstream_init(hipStream_t st, RefStreamState* states, uint32_t seed, int T) {
    hipLaunchKernelGGL(k_refstream_init, dim3((T + 63) / 64), dim3(64), 0, st, states, seed, T);
    return hipGetLastError();
}
hipError_t refstream_discard(hipStream_t st, RefStreamState* states, int t, unsigned long long n) {
    hipLaunchKernelGGL(k_refstream_discard, dim3(1), dim3(RS_THREADS), 0, st, states, t, n);
    return hipGetLastError();
}
size_t refstream_window_bytes(int T, int A) {
    const size_t D = refstream_window_outputs(A);
    return (size_t)T * (D * 4 + (size_t)A * 16 + ((size_t)A + 1) * 4 + (size_t)A * 48 + (size_t)A) + 64;
}
hipError_t refstream_window(hipStream_t st, RefStreamState* states, int T, int A, int mode, void* scratch, const FrameDev& F, int thr_int, const int32_t* first,
                            int32_t* served, int32_t* need, int32_t* parsed, unsigned long long* consumed, long long* attempts, double* poses, int32_t* sets_out,
                            uint8_t* ok, float* staged) {
    const int D = refstream_window_outputs(A);
    char* p = reinterpret_cast<char*>(scratch);
    double* poses_tmp = reinterpret_cast<double*>(p); p += (size_t)T * A * 48;
    uint32_t* raw = reinterpret_cast<uint32_t*>(p); p += (size_t)T * D * 4;
    int32_t* sets = reinterpret_cast<int32_t*>(p); p += (size_t)T * A * 16;
    uint32_t* offs = reinterpret_cast<uint32_t*>(p); p += (size_t)T * (A + 1) * 4;
    uint8_t* ok_tmp = reinterpret_cast<uint8_t*>(p);
    hipLaunchKernelGGL(k_refstream_parse, dim3(T), dim3(RS_THREADS), 0, st, states, F.W, F.H, mode, A, D, raw, sets, offs, parsed, need);
    hipLaunchKernelGGL(k_refstream_eval, dim3((A + 63) / 64, T), dim3(64), 0, st, A, parsed, sets, F, thr_int, poses_tmp, ok_tmp);
    hipLaunchKernelGGL(k_refstream_select, dim3(T), dim3(RS_THREADS), 0, st, states, A, parsed, sets, offs, poses_tmp, ok_tmp, first, served, need, consumed, attempts, F, poses,
                       sets_out, ok, staged);
    return hipGetLastError();
}
hipError_t refstream_unserved(hipStream_t st, int T, const int32_t* first, const int32_t* served, const int32_t* need, const FrameDev& F, double* poses, int32_t* sets_out,
                              uint8_t* ok, float* staged) {
    hipLaunchKernelGGL(k_refstream_unserved, dim3(T), dim3(64), 0, st, T, first, served, need, F, poses, sets_out, ok, staged);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K5: dPNP.  Lane l of hypothesis h: coordinate c = l / 2 (point i = c / 3, axis j = c % 3), sign = l & 1.
// The object points are perturbed in float, sequentially (+eps, -2 eps, +eps) like the reference
// (core/cnn_softam.h:115-135), so a lane first replays the float round trips of the coordinates before
// its own.  32 lanes per hypothesis (24 active), two hypotheses per wave.
// --------------------------------------------------------------------------------------------------
// One workgroup of two waves per hypothesis: 24 solves (coordinate c = 0..11, +/-) x 4 lanes per solve, one per quartic root -- the
// roots' Horn/Jacobi alignments, the long pole of a solve, run side by side as in K1's four-lane form (round 1 ran them in sequence on
// ONE lane per solve with two hypotheses per wave: 62 us for 256 hypotheses on 128 of the chip's 1024 SIMDs).
// Thread t: solve s = t >> 2 (forward s = 2c, backward s = 2c + 1: neighbouring quads), root t & 3.
__global__ __launch_bounds__(128) void k_dpnp(int N, const int32_t* __restrict__ sets, FrameDev F, float eps, double* __restrict__ J, int Nf) {
    const int h = blockIdx.x;
    if (Nf > 0) {  // frame batch: hypothesis h belongs to frame h / Nf
        const int f = h / Nf;
        F.xyz += (long long)f * F.xyz_stride;
        if (F.uv) F.uv += (long long)f * F.uv_stride;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int s = t >> 2, root = t & 3;
    const bool active = s < 24;
    const int c = s >> 1;
    __shared__ int s_nan[2];
    double jp6[6] = {0, 0, 0, 0, 0, 0};
    bool cand = false;
    double Rc[9], Tc[3], reproj = 0;
    if (active) {
        float X[4][3], uv[4][2];
        int32_t set4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) set4[j] = sets[(size_t)h * 4 + j];
        load_set(F, set4, X, uv);
        // replay the float round trips of coordinates 0..c-1, then apply this solve's own perturbation
#pragma unroll
        for (int cc = 0; cc < 12; cc++) {
            float& v = X[cc / 3][cc % 3];
            if (cc < c) { v += eps; v -= 2 * eps; v += eps; }
            else if (cc == c) { v += eps; if (s & 1) v -= 2 * eps; }
        }
        const dm::Cam K = make_cam(F);
        dm::P3PSetup S;
        if (dm::p3p_setup(X, uv, K, S) && root < S.n) {
            const double x = (root == 0) ? S.roots[0] : (root == 1) ? S.roots[1] : (root == 2) ? S.roots[2] : S.roots[3];
            // the least-squares alignment, like OpenCV: the difference quotient amplifies whatever an alignment that is not the optimum (the triad of
            // rounds 1-4: 5e-2) adds.  Closed form since round 5; -DDSAC_K5_ALIGN_JACOBI: OpenCV's own Jacobi sweeps (the long pole of a solve)
#ifdef DSAC_K5_ALIGN_JACOBI
            cand = dm::p3p_eval_root<true>(S, X, uv, K, x, Rc, Tc, reproj);
#else
            cand = dm::p3p_eval_root<false>(S, X, uv, K, x, Rc, Tc, reproj);
#endif
        }
    }
    const int win = dm::best_root_of_quad(cand, reproj);  // -1: no root -> safeSolvePnP's zero pose
    {
        double cv6[6] = {0, 0, 0, 0, 0, 0};
        if (active && win == root) {
            dm::rodrigues_m2v(Rc, cv6);
            cv6[3] = Tc[0]; cv6[4] = Tc[1]; cv6[5] = Tc[2];
        }
        // lanes other than the winner (and all four when there is none) carry the zero pose; the winner's is summed into root lane 0
        double mine[6];
        if (active && (win == root || (win < 0 && root == 0))) dm::cv_to_jp6(cv6, mine);
        else {
#pragma unroll
            for (int k = 0; k < 6; k++) mine[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double v0 = dm::quad_bcast_d<0>(mine[k]), v1 = dm::quad_bcast_d<1>(mine[k]), v2 = dm::quad_bcast_d<2>(mine[k]), v3 = dm::quad_bcast_d<3>(mine[k]);
            jp6[k] = (win <= 0) ? v0 : (win == 1) ? v1 : (win == 2) ? v2 : v3;  // exactly one lane of the quad holds a pose
        }
    }
    // central difference: forward quad minus the backward quad four lanes up
    const double inv = 1.0 / (double)(2 * eps);
    bool nan = false;
    double d[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double other = __shfl_xor(jp6[k], 4, 64);
        d[k] = (jp6[k] - other) * inv;
        nan = nan || (d[k] != d[k]);
    }
    // any NaN in any column -> whole Jacobian zero (core/cnn_softam.h:141-142)
    const bool writer = active && !(s & 1) && root == 0;
    const unsigned long long m = __ballot(nan && writer);
    if (lane == 0) s_nan[t >> 6] = m != 0ull;
    __syncthreads();
    const bool any_nan = s_nan[0] || s_nan[1];
    if (writer) {
#pragma unroll
        for (int k = 0; k < 6; k++) J[(size_t)h * 72 + k * 12 + c] = any_nan ? 0.0 : d[k];
    }
}

hipError_t dpnp(hipStream_t st, int N, const int32_t* sets, const FrameDev& F, float eps, double* J, int Nf) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dpnp, dim3(N), dim3(128), 0, st, N, sets, F, eps, J, F.frames > 1 ? Nf : 0);
    return hipGetLastError();
}

}  // namespace dk

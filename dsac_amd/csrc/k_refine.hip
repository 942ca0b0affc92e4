// k_refine.hip -- K6: inlier refinement (Levenberg-Marquardt PnP) and its finite-difference replicas.
//
// Replaces the refinement loop of processImage (core/cnn_softam.h:1099-1154), the replay helper refine()
// (:663-723) and the central-difference drivers dRefineHyp (:738-836) / dRefineObj (:853-923).
//
// Design: ONE WAVE PER REFINEMENT PROBLEM.  The reference recomputes a full error image before every
// refinement step (getDiffMap, P residuals) although the walk along the pixel permutation stops as soon as
// `max_inl` (100) inliers are found -- typically after a few hundred cells.  Here the wave evaluates the
// residuals lazily, 64 permuted cells at a time (fp64, the arithmetic of getDiffMap), and compacts the
// inliers in permutation order with ballot + popcount, so a refinement step touches O(max_inl) cells
// instead of P.  The LM solve (CvLevMarq's state machine: Marquardt scaling 1+lambda, lambda0 = 1e-3,
// x10 / /10, <= 20 iterations, eps = FLT_EPSILON on the relative parameter change) runs on the same wave:
// lanes split the <= max_inl correspondences, J^T J / J^T e are reduced with a butterfly (bit-identical on
// all lanes), and every lane solves the damped 6x6 system redundantly, which keeps control flow uniform.
// The 12 + 6*n finite-difference replicas of dRefineHyp/dRefineObj are just more waves of the same kernel:
// a replica is (start pose, optionally one replaced coordinate), so the whole Jacobian is one launch.
#include "kernels.h"
#include "dmath.h"

namespace dk {

constexpr int RF_MAX_INL = 256;  // LDS capacity for the collected correspondences

DM_INLINE dm::Cam make_cam_r(const FrameDev& F) { return dm::Cam{(double)F.fx, (double)F.fy, (double)F.cx, (double)F.cy}; }

// ---- wave-wide sums of fp64 values without the LDS crossbar (round 1 used 6 ds_bpermute round trips per value) ----
DM_INLINE void dsplit(double v, unsigned& lo, unsigned& hi) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    lo = (unsigned)u;
    hi = (unsigned)(u >> 32);
}
DM_INLINE double djoin(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
// swap(a, b): the upper 32 lanes (odd 16-rows) of a trade places with the lower 32 lanes (even rows) of b
DM_INLINE void dswap32(double& a, double& b) {
    unsigned al, ah, bl, bh;
    dsplit(a, al, ah);
    dsplit(b, bl, bh);
    auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    a = djoin(l[0], h[0]);
    b = djoin(l[1], h[1]);
}
DM_INLINE void dswap16(double& a, double& b) {
    unsigned al, ah, bl, bh;
    dsplit(a, al, ah);
    dsplit(b, bl, bh);
    auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
    a = djoin(l[0], h[0]);
    b = djoin(l[1], h[1]);
}
template <int CTRL, int BANK = 0xf>
DM_INLINE double ddpp(double old, double src) {
    unsigned ol, oh, sl, sh;
    dsplit(old, ol, oh);
    dsplit(src, sl, sh);
    const unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)ol, (int)sl, CTRL, 0xf, BANK, false);
    const unsigned h = (unsigned)__builtin_amdgcn_update_dpp((int)oh, (int)sh, CTRL, 0xf, BANK, false);
    return djoin(l, h);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128, DPP_SHL4 = 0x104, DPP_SHR4 = 0x114;
// value of lane (l ^ 4): lanes with bit 2 clear read from l + 4, the others from l - 4 (bank masks pick the lanes that take each move)
DM_INLINE double dxor4(double v) { return ddpp<DPP_SHR4, 0xA>(ddpp<DPP_SHL4, 0x5>(v, v), v); }

DM_INLINE void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one value, result in every lane (the same bits everywhere: each stage adds the same two partial sums in both partners)
DM_INLINE double wave_allsum(double v) {
    v += ddpp<DPP_XOR1>(v, v);
    v += ddpp<DPP_XOR2>(v, v);
    v += ddpp<DPP_HALF_MIRROR>(v, v);
    v += ddpp<DPP_MIRROR>(v, v);  // every lane of a 16-row holds the row's sum
    double a = v, b = v;
    dswap16(a, b);
    v = a + b;
    a = v;
    b = v;
    dswap32(a, b);
    return a + b;
}

// 32 values per lane -> their 32 wave totals in s_out[0..31] (LDS), readable by every lane after the call.  Transpose-reduce: each
// stage halves the values a lane still carries and the lanes that remain to be summed (31 adds and 62 lane moves in all instead of
// 32 x 6 adds and 32 x 12 moves).  After stage k a lane holds the values whose index bits agree with its own lane bits:
//   index = lane bit3 + 2 * bit1 + 4 * bit0 + 8 * bit4 + 16 * bit5;   lanes l and l ^ 4 end with the same index (summed last).
DM_INLINE void wave_allsum32(double (&v)[32], double* s_out) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 16; i++) { dswap32(v[i], v[i + 16]); v[i] += v[i + 16]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { dswap16(v[i], v[i + 8]); v[i] += v[i + 8]; }
    const bool b0 = lane & 1, b1 = lane & 2, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; i++) {  // partner l ^ 1: keep v[i] (bit0 clear) or v[i + 4], send the other one
        const double keep = b0 ? v[i + 4] : v[i], send = b0 ? v[i] : v[i + 4];
        v[i] = keep + ddpp<DPP_XOR1>(send, send);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double keep = b1 ? v[i + 2] : v[i], send = b1 ? v[i] : v[i + 2];
        v[i] = keep + ddpp<DPP_XOR2>(send, send);
    }
    const double keep = b3 ? v[1] : v[0], send = b3 ? v[0] : v[1];
    double r = keep + ddpp<DPP_ROR8>(send, send);
    r += dxor4(r);
    const int idx = ((lane >> 3) & 1) + 2 * ((lane >> 1) & 1) + 4 * (lane & 1) + 8 * ((lane >> 4) & 1) + 16 * ((lane >> 5) & 1);
    // one wave per workgroup: its LDS operations execute in order, so a wave barrier (no counter wait: global loads may stay in
    // flight) is all that separates the readers of the previous result, this write, and the readers of this one
    wave_sync_lds();
    s_out[idx] = r;
    wave_sync_lds();
}

// Solve of the symmetric positive definite 6x6 system A x = b (upper triangle of A, row-major 6x6) by L D L^T: six reciprocals and no
// square root on the dependent chain (a Cholesky factorisation has 6 square roots and 27 divisions there).  OpenCV uses an SVD
// pseudo-inverse here; identical for full-rank normal equations.
DM_INLINE bool solve6_spd(const double A[36], const double b[6], double x[6]) {
    double L[36], W[36], inv[6];  // W[i][j] = L[i][j] * D[j]
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * W[j * 6 + k];
        ok = ok && (d > 0.0);
        inv[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double t = A[j * 6 + i];
#pragma unroll
            for (int k = 0; k < j; k++) t -= L[i * 6 + k] * W[j * 6 + k];
            W[i * 6 + j] = t;
            L[i * 6 + j] = t * inv[j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double t = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) t -= L[i * 6 + k] * y[k];
        y[i] = t;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double t = y[i] * inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) t -= L[k * 6 + i] * x[k];
        x[i] = t;
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = 0.0;
    }
    return ok;
}

// Residuals (and optionally the normal equations) of the n collected correspondences at pose `p`.
// Returns the L2 norm of the 2n residuals; all lanes return the same bits.
template <bool WITH_J>
DM_INLINE double lm_eval(int n, const float* s_X, const float* s_uv, double* s_red, const dm::Cam& K, const double p[6], double JtJ[21], double JtE[6]) {
    const int lane = threadIdx.x & 63;
    double R[9], dRdr[27];
    dm::rodrigues_v2m<WITH_J>(p, R, dRdr);
    double acc[32];  // 21 of J^T J, 6 of J^T e, |e|^2, 4 unused
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double Mx = s_X[i * 3], My = s_X[i * 3 + 1], Mz = s_X[i * 3 + 2];
        const double Xc = R[0] * Mx + R[1] * My + R[2] * Mz + p[3];
        const double Yc = R[3] * Mx + R[4] * My + R[5] * Mz + p[4];
        const double Zc = R[6] * Mx + R[7] * My + R[8] * Mz + p[5];
        const double z = (Zc != 0.0) ? 1. / Zc : 1.;
        const double x = Xc * z, y = Yc * z;
        const double eu = x * K.fx + K.cx - (double)s_uv[i * 2];
        const double ev = y * K.fy + K.cy - (double)s_uv[i * 2 + 1];
        acc[27] += eu * eu + ev * ev;
        if (WITH_J) {
            double Ju[6], Jv[6];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double* dR = dRdr + j * 9;
                const double dX = dR[0] * Mx + dR[1] * My + dR[2] * Mz;
                const double dY = dR[3] * Mx + dR[4] * My + dR[5] * Mz;
                const double dZ = dR[6] * Mx + dR[7] * My + dR[8] * Mz;
                Ju[j] = K.fx * z * (dX - x * dZ);
                Jv[j] = K.fy * z * (dY - y * dZ);
            }
            Ju[3] = K.fx * z; Ju[4] = 0; Ju[5] = -K.fx * x * z;
            Jv[3] = 0; Jv[4] = K.fy * z; Jv[5] = -K.fy * y * z;
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                acc[21 + a] += Ju[a] * eu + Jv[a] * ev;
#pragma unroll
                for (int b2 = a; b2 < 6; b2++) { acc[q] += Ju[a] * Ju[b2] + Jv[a] * Jv[b2]; q++; }
            }
        }
    }
    if (!WITH_J) return sqrt(wave_allsum(acc[27]));
    wave_allsum32(acc, s_red);
#pragma unroll
    for (int i = 0; i < 21; i++) JtJ[i] = s_red[i];
#pragma unroll
    for (int i = 0; i < 6; i++) JtE[i] = s_red[21 + i];
    const double e2 = s_red[27];
    return sqrt(e2);
}

__constant__ double c_pow10[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0,
                                   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};

DM_INLINE void lm_step(const double JtJ[21], const double JtE[6], int lambdaLg10, const double prev[6], double param[6]) {
    const double lambda = c_pow10[lambdaLg10 + 16];  // CvLevMarq: exp(lambdaLg10 * log(10)), lambdaLg10 in [-16, 16]
    double A[36];
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) { A[a * 6 + b] = JtJ[q]; A[b * 6 + a] = JtJ[q]; q++; }
#pragma unroll
    for (int i = 0; i < 6; i++) A[i * 7] *= 1. + lambda;
    double dx[6];
    solve6_spd(A, JtE, dx);
#pragma unroll
    for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
}

// solvePnP(CV_ITERATIVE, useExtrinsicGuess = true) on the wave; pose is updated in place.
DM_INLINE void lm_pnp(int n, const float* s_X, const float* s_uv, double* s_red, const dm::Cam& K, double pose[6]) {
    double param[6], prev[6], JtJ[21], JtE[6];
#pragma unroll
    for (int i = 0; i < 6; i++) param[i] = pose[i];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = 1.7976931348623157e308, errNorm = 0;
    double e_at_param = lm_eval<true>(n, s_X, s_uv, s_red, K, param, JtJ, JtE);
    bool done = false;
    for (int guard = 0; guard < 64 && !done; guard++) {
#pragma unroll
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        if (iters == 0) prevErrNorm = e_at_param;
        lm_step(JtJ, JtE, lambdaLg10, prev, param);
        for (int inner = 0; inner < 40; inner++) {
            errNorm = lm_eval<false>(n, s_X, s_uv, s_red, K, param, nullptr, nullptr);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { lm_step(JtJ, JtE, lambdaLg10, prev, param); continue; }
            }
            lambdaLg10 = max(lambdaLg10 - 1, -16);
            double num = 0, den = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) { num += (param[i] - prev[i]) * (param[i] - prev[i]); den += prev[i] * prev[i]; }
            const double change = sqrt(num) / (sqrt(den) + 2.220446049250313e-16);
            if (++iters >= 20 || change < 1.1920928955078125e-07) { done = true; break; }
            prevErrNorm = errNorm;
            e_at_param = lm_eval<true>(n, s_X, s_uv, s_red, K, param, JtJ, JtE);
            break;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) pose[i] = param[i];
}

// The inlier walk reads the permuted cells 64 * WALK_AHEAD at a time: permutation entry -> coordinate is a chain of two dependent
// loads, so one round trip now serves 256 cells (a walk that collects max_inl = 100 inliers rarely needs more).
constexpr int WALK_AHEAD = 4;
struct WalkCells {
    int p[WALK_AHEAD];  // cell index, -1 past the end of the permutation
    float X[WALK_AHEAD], Y[WALK_AHEAD], Z[WALK_AHEAD], u[WALK_AHEAD], v[WALK_AHEAD];
};
DM_INLINE void load_walk_cells(const int32_t* __restrict__ pidx, int base, int lane, const FrameDev& F, WalkCells& c) {
    const int P = F.P;
    int q[WALK_AHEAD];
#pragma unroll
    for (int s = 0; s < WALK_AHEAD; s++) {
        const int idx = base + s * 64 + lane;
        q[s] = idx < P ? min(max(pidx[idx], 0), P - 1) : -1;
    }
#pragma unroll
    for (int s = 0; s < WALK_AHEAD; s++) {
        const int p = max(q[s], 0);
        c.p[s] = q[s];
        c.X[s] = F.xyz[(size_t)p * 3];
        c.Y[s] = F.xyz[(size_t)p * 3 + 1];
        c.Z[s] = F.xyz[(size_t)p * 3 + 2];
        if (F.uv) { c.u[s] = F.uv[(size_t)p * 2]; c.v[s] = F.uv[(size_t)p * 2 + 1]; }
        else { const int y = p / F.W; c.u[s] = (float)(p - y * F.W); c.v[s] = (float)y; }
    }
}

__global__ __launch_bounds__(64) void k_refine(int B, const int32_t* __restrict__ n_live, int live_base, int live_mul,
                                               const double* __restrict__ init_poses, const int32_t* __restrict__ perm, int steps, int max_inl,
                                               int min_inl, float thr, const int32_t* __restrict__ pert_px_c, const float* __restrict__ pert_value,
                                               FrameDev F, double* __restrict__ out_poses, int32_t* __restrict__ inlier_map,
                                               int32_t* __restrict__ steps_done, int map_stride, int group) {
    const int b = blockIdx.x;
    if (b >= B) return;
    // replica lists shorter than the launch: `group` replicas per list (0: one list), list m holds live_base + live_mul * n_live[m]
    if (n_live) {
        const int m = group > 0 ? b / group : 0, local = group > 0 ? b - m * group : b;
        if (local >= live_base + live_mul * n_live[m]) return;
    }
    const int lane = threadIdx.x;
    __shared__ float s_X[RF_MAX_INL * 3];
    __shared__ float s_uv[RF_MAX_INL * 2];
    __shared__ double s_red[32];
    const dm::Cam K = make_cam_r(F);
    double pose[6];
#pragma unroll
    for (int i = 0; i < 6; i++) pose[i] = init_poses[(size_t)b * 6 + i];
    const int ppx = pert_px_c ? pert_px_c[2 * b] : -1;
    const int pch = pert_px_c ? pert_px_c[2 * b + 1] : 0;
    const float pval = pert_value ? pert_value[b] : 0.f;
    const int P = F.P;
    int done = 0;
    WalkCells cells;
    if (steps > 0) load_walk_cells(perm, 0, lane, F, cells);
    for (int step = 0; step < steps; step++) {
        double R[9];
        dm::rodrigues_v2m<false>(pose, R, nullptr);
        const int32_t* pidx = perm + (size_t)step * P;
        int cnt = 0;
        for (int base = 0; base < P && cnt < max_inl; base += 64 * WALK_AHEAD) {
            if (base > 0) load_walk_cells(pidx, base, lane, F, cells);
#pragma unroll
            for (int s = 0; s < WALK_AHEAD; s++) {
                if (cnt >= max_inl) break;  // uniform: the walk stops at max_inl taken cells (core/cnn_softam.h:1121-1135)
                const int p = cells.p[s];
                const bool in = p >= 0;
                float X = cells.X[s], Y = cells.Y[s], Z = cells.Z[s];
                if (p == ppx) { if (pch == 0) X = pval; else if (pch == 1) Y = pval; else Z = pval; }
                const float pu = cells.u[s], pv = cells.v[s];
                const float e = dm::residual_f(R, pose + 3, K, X, Y, Z, pu, pv, 100.0);
                const bool inl = in && (e < thr);
                const unsigned long long m = __ballot(inl);
                const int prefix = __popcll(m & ((1ull << lane) - 1ull));
                const bool take = inl && (cnt + prefix < max_inl);
                if (take) {
                    const int slot = cnt + prefix;
                    s_X[slot * 3] = X; s_X[slot * 3 + 1] = Y; s_X[slot * 3 + 2] = Z;
                    s_uv[slot * 2] = pu; s_uv[slot * 2 + 1] = pv;
                    if (inlier_map && (map_stride > 0 || b == 0)) atomicAdd(&inlier_map[(size_t)b * map_stride + p], 1);
                }
                cnt += __popcll(m);
            }
        }
        // the head of the next step's walk does not depend on the pose: its loads fly under the LM solve
        if (step + 1 < steps) load_walk_cells(pidx + P, 0, lane, F, cells);
        wave_sync_lds();
        const int n = min(cnt, max_inl);
        if (n < min_inl) break;  // abort for stability: too few inliers (core/cnn_softam.h:700, 1136)
        double upd[6];
#pragma unroll
        for (int i = 0; i < 6; i++) upd[i] = pose[i];
        lm_pnp(n, s_X, s_uv, s_red, K, upd);
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 6; i++) nan = nan || (upd[i] != upd[i]);
        if (nan) break;
#pragma unroll
        for (int i = 0; i < 6; i++) pose[i] = upd[i];
        done++;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) out_poses[(size_t)b * 6 + i] = pose[i];
        if (steps_done) steps_done[b] = done;
    }
}

hipError_t refine(hipStream_t st, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                  const int32_t* pert_px_c, const float* pert_value, const FrameDev& F, double* out_poses, int32_t* inlier_map,
                  int32_t* steps_done, int map_stride) {
    if (B <= 0) return hipSuccess;
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_refine, dim3(B), dim3(64), 0, st, B, (const int32_t*)nullptr, 0, 0, init_poses, perm, steps, max_inl, min_inl, thr, pert_px_c,
                       pert_value, F, out_poses, inlier_map, steps_done, map_stride, 0);
    return hipGetLastError();
}

__global__ void k_zero_set_cells(int N, const int32_t* __restrict__ sets, int P, int32_t* __restrict__ maps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 4) return;
    const int h = i >> 2;
    const int p = sets[i];
    if (p >= 0 && p < P) maps[(size_t)h * P + p] = 0;
}
hipError_t zero_set_cells(hipStream_t st, int N, const int32_t* sets, int P, int32_t* inlier_maps) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_set_cells, dim3((N * 4 + 255) / 256), dim3(256), 0, st, N, sets, P, inlier_maps);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// Replica plan of dRefineHyp (12 replicas) + dRefineObj (6 per selected cell).  One wave scans inlier_map in
// the reference's x-outer / y-inner order (core/cnn_softam.h:873-882) and keeps every skip-th inlier cell.
// --------------------------------------------------------------------------------------------------
// Parallel form of the column-major scan: PLAN_THREADS threads own consecutive segments of the scan order t = x * H + y, count their
// inlier cells, an LDS prefix sum turns the counts into each segment's starting inCount, and a second walk selects: the k-th inlier
// cell (1-based) is kept iff k % skip == 0 and then is selection number k / skip - 1 -- a closed form, so no second prefix is needed.
// (Round 1 scanned with one wave, 64 cells per dependent step: 1.4 ms on a 640 x 480 map.)
constexpr int PLAN_THREADS = 1024;

DM_INLINE int plan_segment_prefix(const int32_t* __restrict__ inlier_map, const FrameDev& F, int t0, int t1, int* s_cnt, int* total) {
    const int tid = threadIdx.x;
    int cnt = 0;
    for (int t = t0; t < t1; t++) { const int x = t / F.H, y = t - x * F.H; cnt += inlier_map[y * F.W + x] != 0; }
    s_cnt[tid] = cnt;
    __syncthreads();
    // Hillis-Steele inclusive scan over PLAN_THREADS counters
    for (int o = 1; o < PLAN_THREADS; o <<= 1) {
        const int v = (tid >= o) ? s_cnt[tid - o] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    *total = s_cnt[PLAN_THREADS - 1];
    return s_cnt[tid] - cnt;  // exclusive prefix = inCount before this segment
}

__global__ __launch_bounds__(PLAN_THREADS) void k_refine_fd_plan(const double* __restrict__ init_pose, const int32_t* __restrict__ inlier_map, FrameDev F,
                                                                 int skip, float eps_hyp, float eps_obj, int cap, double* __restrict__ rep_poses,
                                                                 int32_t* __restrict__ rep_px_c, float* __restrict__ rep_value,
                                                                 int32_t* __restrict__ obj_pixels, int32_t* __restrict__ n_obj) {
    __shared__ int s_cnt[PLAN_THREADS];
    const int lane = threadIdx.x;
    double init[6];
#pragma unroll
    for (int i = 0; i < 6; i++) init[i] = init_pose[i];
    // dRefineHyp: replica 2i = +step on parameter i, 2i+1 = (+step) - 2 step  (double arithmetic, :758-772,798-812)
    if (lane < 12) {
        const int i = lane >> 1;
        const double step = (i < 3) ? (double)eps_hyp : (double)(eps_hyp * 1000);
        double pose[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pose[k] = init[k];
        double v = init[i] + step;
        if (lane & 1) v -= 2 * step;
#pragma unroll
        for (int k = 0; k < 6; k++) rep_poses[(size_t)lane * 6 + k] = (k == i) ? v : pose[k];
        rep_px_c[2 * lane] = -1; rep_px_c[2 * lane + 1] = 0; rep_value[lane] = 0.f;
    }
    // dRefineObj: column-major scan
    const int P = F.P;
    const int seg = (P + PLAN_THREADS - 1) / PLAN_THREADS;
    const int t0 = min(P, lane * seg), t1 = min(P, t0 + seg);
    int total = 0;
    int inCount = plan_segment_prefix(inlier_map, F, t0, t1, s_cnt, &total);
    for (int t = t0; t < t1; t++) {
        const int x = t / F.H, y = t - x * F.H, p = y * F.W + x;
        if (inlier_map[p] == 0) continue;
        inCount++;
        if (inCount % skip != 0) continue;
        const int slot = inCount / skip - 1;
        if (slot >= cap) continue;
        obj_pixels[slot] = p;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v0 = F.xyz[(size_t)p * 3 + c];
            const float vf = v0 + eps_obj;
            const float vb = vf - 2 * eps_obj;
            const int r = 12 + slot * 6 + c * 2;
#pragma unroll
            for (int k = 0; k < 6; k++) { rep_poses[(size_t)r * 6 + k] = init[k]; rep_poses[(size_t)(r + 1) * 6 + k] = init[k]; }
            rep_px_c[2 * r] = p; rep_px_c[2 * r + 1] = c; rep_value[r] = vf;
            rep_px_c[2 * (r + 1)] = p; rep_px_c[2 * (r + 1) + 1] = c; rep_value[r + 1] = vb;
        }
    }
    if (lane == 0) n_obj[0] = min(total / skip, cap);
}

hipError_t refine_fd_plan(hipStream_t st, const double* init_pose, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_hyp,
                          float eps_obj, int cap, double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels, int32_t* n_obj) {
    hipLaunchKernelGGL(k_refine_fd_plan, dim3(1), dim3(PLAN_THREADS), 0, st, init_pose, inlier_map, F, skip, eps_hyp, eps_obj, cap, rep_poses, rep_px_c, rep_value,
                       obj_pixels, n_obj);
    return hipGetLastError();
}

// launches the replicas: grid = 12 + 6*cap waves, those beyond 12 + 6*n_obj exit immediately
hipError_t refine_fd_run(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                         int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out) {
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    const int B = 12 + 6 * cap;
    hipLaunchKernelGGL(k_refine, dim3(B), dim3(64), 0, st, B, n_obj, 12, 6, rep_poses, perm, steps, max_inl, min_inl, thr, rep_px_c, rep_value, F, rep_out,
                       (int32_t*)nullptr, (int32_t*)nullptr, 0, 0);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// DSAC variant (core/cnn.h:854-990 dRefine): the refinement restarts from P3P of the minimal set (cnn.h:797-800), so the
// first three set points are perturbed too (the 4th only disambiguates: "gradient is anyway zero", :872).  Replicas
// 0..17 = (point pt, channel c, +/-), then 6 per selected inlier cell exactly as in dRefineObj.  The start pose of every
// replica is P3P of the set read from the replica's perturbed map -- for the inlier replicas that is the unperturbed
// hypothesis, because processImage removes the set's own cells from the inlier map (:1208-1214).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PLAN_THREADS) void k_refine_fd_plan_set(const int32_t* __restrict__ set4, const int32_t* __restrict__ inlier_map, FrameDev F, int skip,
                                                           float eps_obj, int cap, int32_t* __restrict__ rep_px_c, float* __restrict__ rep_value,
                                                           int32_t* __restrict__ obj_pixels, int32_t* __restrict__ n_obj) {
    const int lane = threadIdx.x;
    const int P = F.P;
    {   // hypothesis m of a batch (blockIdx.y): its set, inlier map and slice of the replica arrays (18 + 6*cap replicas each)
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        set4 += 4 * m; inlier_map += m * P; rep_px_c += m * R * 2; rep_value += m * R; obj_pixels += m * cap; n_obj += m;
    }
    if (lane < 18) {
        const int pt = lane / 6, c = (lane % 6) >> 1;
        const int p = min(max(set4[pt], 0), P - 1);
        const float v0 = F.xyz[(size_t)p * 3 + c];
        const float vf = v0 + eps_obj;
        rep_px_c[2 * lane] = p; rep_px_c[2 * lane + 1] = c;
        rep_value[lane] = (lane & 1) ? vf - 2 * eps_obj : vf;
    }
    __shared__ int s_cnt[PLAN_THREADS];
    const int seg = (P + PLAN_THREADS - 1) / PLAN_THREADS;  // x-outer / y-inner order: t = x * H + y  (cnn.h:935-945)
    const int t0 = min(P, lane * seg), t1 = min(P, t0 + seg);
    int total = 0;
    int inCount = plan_segment_prefix(inlier_map, F, t0, t1, s_cnt, &total);
    for (int t = t0; t < t1; t++) {
        const int x = t / F.H, y = t - x * F.H, p = y * F.W + x;
        if (inlier_map[p] == 0) continue;
        inCount++;
        if (inCount % skip != 0) continue;
        const int slot = inCount / skip - 1;
        if (slot >= cap) continue;
        obj_pixels[slot] = p;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v0 = F.xyz[(size_t)p * 3 + c];
            const float vf = v0 + eps_obj;
            const int r = 18 + slot * 6 + c * 2;
            rep_px_c[2 * r] = p; rep_px_c[2 * r + 1] = c; rep_value[r] = vf;
            rep_px_c[2 * (r + 1)] = p; rep_px_c[2 * (r + 1) + 1] = c; rep_value[r + 1] = vf - 2 * eps_obj;
        }
    }
    if (lane == 0) n_obj[0] = min(total / skip, cap);
}

// start pose of replica r: P3P (Horn alignment, as OpenCV) of the set read through the replica's perturbation
__global__ __launch_bounds__(64) void k_refine_fd_init_set(int cap, const int32_t* __restrict__ n_obj, const int32_t* __restrict__ set4,
                                                           const int32_t* __restrict__ rep_px_c, const float* __restrict__ rep_value, FrameDev F,
                                                           double* __restrict__ rep_poses) {
    {
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        set4 += 4 * m; rep_px_c += m * R * 2; rep_value += m * R; rep_poses += m * R * 6; n_obj += m;
    }
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= 18 + 6 * min(n_obj[0], cap)) return;
    const int ppx = rep_px_c[2 * r], pch = rep_px_c[2 * r + 1];
    const float pval = rep_value[r];
    float X[4][3], uv[4][2];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = min(max(set4[j], 0), F.P - 1);
#pragma unroll
        for (int c = 0; c < 3; c++) X[j][c] = (p == ppx && c == pch) ? pval : F.xyz[(size_t)p * 3 + c];
        if (F.uv) { uv[j][0] = F.uv[(size_t)p * 2]; uv[j][1] = F.uv[(size_t)p * 2 + 1]; }
        else { const int y = p / F.W; uv[j][0] = (float)(p - y * F.W); uv[j][1] = (float)y; }
    }
    double cv6[6];
    if (!dm::p3p<true>(X, uv, make_cam_r(F), cv6)) {
#pragma unroll
        for (int k = 0; k < 6; k++) cv6[k] = 0;  // safeSolvePnP's zero pose
    }
#pragma unroll
    for (int k = 0; k < 6; k++) rep_poses[(size_t)r * 6 + k] = cv6[k];
}

hipError_t refine_fd_plan_set(hipStream_t st, const int32_t* set4, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_obj, int cap,
                              double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels, int32_t* n_obj, int M) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_refine_fd_plan_set, dim3(1, M), dim3(PLAN_THREADS), 0, st, set4, inlier_map, F, skip, eps_obj, cap, rep_px_c, rep_value, obj_pixels, n_obj);
    const int R = 18 + 6 * cap;
    hipLaunchKernelGGL(k_refine_fd_init_set, dim3((R + 63) / 64, M), dim3(64), 0, st, cap, n_obj, set4, rep_px_c, rep_value, F, rep_poses);
    return hipGetLastError();
}

hipError_t refine_fd_run_set(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                             int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out, int M) {
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    const int R = 18 + 6 * cap;
    const long long B = (long long)R * M;
    if (B > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_refine, dim3((unsigned)B), dim3(64), 0, st, (int)B, n_obj, 18, 6, rep_poses, perm, steps, max_inl, min_inl, thr, rep_px_c, rep_value, F,
                       rep_out, (int32_t*)nullptr, (int32_t*)nullptr, 0, R);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void k_refine_fd_finish_set(const double* __restrict__ rep_out, const int32_t* __restrict__ n_obj, int cap, int skip,
                                                             float eps_obj, double* __restrict__ J_set, double* __restrict__ J_obj) {
    {
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        rep_out += m * R * 6; n_obj += m; J_set += m * 54; J_obj += m * (size_t)cap * 18;
    }
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;  // 0..8: set point pt = pair / 3, channel pair % 3; then 3 per cell
    const int npairs = 9 + 3 * min(n_obj[0], cap);
    if (pair >= npairs) return;
    double f6[6], b6[6], cvf[6], cvb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { cvf[k] = rep_out[(size_t)(2 * pair) * 6 + k]; cvb[k] = rep_out[(size_t)(2 * pair + 1) * 6 + k]; }
    dm::cv_to_jp6(cvf, f6);
    dm::cv_to_jp6(cvb, b6);
    if (pair < 9) {
#pragma unroll
        for (int k = 0; k < 6; k++) J_set[k * 9 + pair] = (f6[k] - b6[k]) / (double)(2 * eps_obj);  // no skip factor (cnn.h:923)
    } else {
        const int cell = (pair - 9) / 3, c = (pair - 9) % 3;
#pragma unroll
        for (int k = 0; k < 6; k++) J_obj[((size_t)cell * 6 + k) * 3 + c] = (f6[k] - b6[k]) / (double)(2 * eps_obj) * skip;
    }
}

hipError_t refine_fd_finish_set(hipStream_t st, const double* rep_out, const int32_t* n_obj, int cap, int skip, float eps_obj, double* J_set, double* J_obj,
                                int M) {
    if (M <= 0) return hipSuccess;
    const int pairs = 9 + 3 * cap;
    hipLaunchKernelGGL(k_refine_fd_finish_set, dim3((pairs + 63) / 64, M), dim3(64), 0, st, rep_out, n_obj, cap, skip, eps_obj, J_set, J_obj);
    return hipGetLastError();
}

// central differences of the jp 6-vectors (getRodVecAndTrans(Hypothesis(cv2our(.))), core/cnn_softam.h:721-722)
__global__ __launch_bounds__(64) void k_refine_fd_finish(const double* __restrict__ rep_out, const int32_t* __restrict__ n_obj, int cap, int skip,
                                                         float eps_hyp, float eps_obj, double* __restrict__ J_hyp, double* __restrict__ J_obj) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;  // replica pair index: 0..5 hyp, 6.. obj (3 per cell)
    const int npairs = 6 + 3 * min(n_obj[0], cap);
    if (pair >= npairs) return;
    double f6[6], b6[6], cvf[6], cvb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { cvf[k] = rep_out[(size_t)(2 * pair) * 6 + k]; cvb[k] = rep_out[(size_t)(2 * pair + 1) * 6 + k]; }
    dm::cv_to_jp6(cvf, f6);
    dm::cv_to_jp6(cvb, b6);
    if (pair < 6) {
        const int i = pair;
#pragma unroll
        for (int k = 0; k < 3; k++) J_hyp[k * 6 + i] = (f6[k] - b6[k]) / (double)(2 * eps_hyp);
#pragma unroll
        for (int k = 3; k < 6; k++) J_hyp[k * 6 + i] = (f6[k] - b6[k]) / (double)(2 * eps_hyp * 1000);
    } else {
        const int cell = (pair - 6) / 3, c = (pair - 6) % 3;
#pragma unroll
        for (int k = 0; k < 6; k++) J_obj[((size_t)cell * 6 + k) * 3 + c] = (f6[k] - b6[k]) / (double)(2 * eps_obj) * skip;
    }
}

hipError_t refine_fd_finish(hipStream_t st, const double* rep_out, const int32_t* n_obj, int cap, int skip, float eps_hyp, float eps_obj,
                            double* J_hyp, double* J_obj) {
    const int pairs = 6 + 3 * cap;
    hipLaunchKernelGGL(k_refine_fd_finish, dim3((pairs + 63) / 64), dim3(64), 0, st, rep_out, n_obj, cap, skip, eps_hyp, eps_obj, J_hyp, J_obj);
    return hipGetLastError();
}

}  // namespace dk

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
// lanes split the <= max_inl correspondences, J^T J / J^T e are summed over the wave by a transpose-reduce
// (v_permlane32/16_swap + DPP, the totals broadcast through LDS: the same bits in all lanes), and every lane
// solves the damped 6x6 system redundantly (L D L^T), which keeps control flow uniform.
// The kernel is a latency chain on one wave (8 steps x ~3 LM iterations), so round 2 shortened the chain rather
// than widening it: the walk's two dependent loads are issued 256 cells at a time and the next step's head is
// fetched under the LM solve; R(r) and its intermediates are kept between evaluations of the same pose; an
// accepted trial's residual and normal equations come from one pass (profiles/r02_k6_*.txt: 175 -> 90 us).
// The 12 + 6*n finite-difference replicas of dRefineHyp/dRefineObj are just more waves of the same kernel:
// a replica is (start pose, optionally one replaced coordinate), so the whole Jacobian is one launch.
#include "kernels.h"
#include "dmath.h"
#include "loss_math.h"

namespace dk {

constexpr int RF_MAX_INL = 256;  // LDS capacity for the collected correspondences

// a frame index that came from device memory (frame_of of the DSAC-variant batch calls): never trusted beyond the batch
DM_INLINE int clamp_frame(int f, const FrameDev& F) { return min(max(f, 0), max(F.frames, 1) - 1); }
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
// 1 / d for the pivots: v_rcp_f64 and two Newton steps (relative error ~1e-16, half the dependent latency of the IEEE division sequence -- six of them sit
// on the solve's critical path); the oracle solves the same system by Gaussian elimination, the agreement is 1e-13 either way
DM_INLINE double drcp(double d) {
#ifdef DSAC_K6_IEEE_DIV
    return 1.0 / d;
#else
    const double r0 = __builtin_amdgcn_rcp(d);
    double r = fma(fma(-d, r0, 1.0), r0, r0);
    r = fma(fma(-d, r, 1.0), r, r);
    // the IEEE edge cases of 1.0 / d: d = 0 or a denormal whose reciprocal is inf would turn into NaN in the Newton steps (0 x inf); keep the
    // hardware's inf / 0 there, so that degenerate inputs take the branches they took with the division
    return (r == r) ? r : r0;
#endif
}
DM_INLINE bool solve6_spd(const double A[36], const double b[6], double x[6]) {
    double L[36], W[36], inv[6];  // W[i][j] = L[i][j] * D[j]
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * W[j * 6 + k];
        ok = ok && (d > 0.0);
        inv[j] = drcp(d);
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

// R(r) of the last rotation vector the wave evaluated, with the intermediates its derivative needs: the LM state machine evaluates the same
// pose several times in a row (the walk, then the first Jacobian, at the start pose; residual-only, then Jacobian, at an accepted step)
struct RodCache {
    double r[3] = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    double R[9];
    dm::RodAux aux;
};
DM_INLINE void rod_at(RodCache& rc, const double p[6]) {
    if (p[0] == rc.r[0] && p[1] == rc.r[1] && p[2] == rc.r[2]) return;
    rc.r[0] = p[0]; rc.r[1] = p[1]; rc.r[2] = p[2];
    dm::rodrigues_R(p, rc.R, rc.aux);
}

// Residuals (and optionally the normal equations) of the n collected correspondences at pose `p`.
// Returns the L2 norm of the 2n residuals; all lanes return the same bits, and the same bits with or without the normal equations.
template <bool WITH_J>
DM_INLINE double lm_eval(int n, const float* s_X, const float* s_uv, double* s_red, const dm::Cam& K, RodCache& rc, const double p[6], double JtJ[21],
                         double JtE[6]) {
    const int lane = threadIdx.x & 63;
    double dRdr[27];
    rod_at(rc, p);
    if (WITH_J) dm::rodrigues_J(rc.aux, dRdr);
    const double* R = rc.R;
    double acc[32];  // 21 of J^T J, 6 of J^T e, 5 unused
    double e2 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = 0.0;
    // two correspondences per lane and trip (lane, lane + 64: the usual 100 inliers are one trip), written as one block so that the two
    // dependent chains (the division, the Jacobian rows) interleave; a lane without a second correspondence adds exact zeros
    for (int i0 = 0; i0 < n; i0 += 128) {
        double eu[2], ev[2], x[2], y[2], zf[2], M[2][3];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = i0 + h * 64 + lane;
            const bool valid = i < n;
            const int ic = valid ? i : 0;
            M[h][0] = s_X[ic * 3]; M[h][1] = s_X[ic * 3 + 1]; M[h][2] = s_X[ic * 3 + 2];
            const double Xc = R[0] * M[h][0] + R[1] * M[h][1] + R[2] * M[h][2] + p[3];
            const double Yc = R[3] * M[h][0] + R[4] * M[h][1] + R[5] * M[h][2] + p[4];
            const double Zc = R[6] * M[h][0] + R[7] * M[h][1] + R[8] * M[h][2] + p[5];
            const double z = (Zc != 0.0) ? drcp(Zc) : 1.;
            const double xx = Xc * z, yy = Yc * z;
            const double du = xx * K.fx + K.cx - (double)s_uv[ic * 2];
            const double dv = yy * K.fy + K.cy - (double)s_uv[ic * 2 + 1];
            x[h] = valid ? xx : 0.0;
            y[h] = valid ? yy : 0.0;
            eu[h] = valid ? du : 0.0;
            ev[h] = valid ? dv : 0.0;
            zf[h] = valid ? z : 0.0;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            e2 += eu[h] * eu[h] + ev[h] * ev[h];
            if (WITH_J) {
                const double Mx = M[h][0], My = M[h][1], Mz = M[h][2], z = zf[h];
                // rows of the 2 x 6 Jacobian: u-row (ju0 ju1 ju2 tu 0 wu), v-row (jv0 jv1 jv2 0 tv wv); the two structural zeros
                // take 23 of the 54 products out of J^T J / J^T e (entry (3,4) is identically zero)
                double ju[3], jv[3];
                const double tu = K.fx * z, tv = K.fy * z;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const double* dR = dRdr + j * 9;
                    const double dX = dR[0] * Mx + dR[1] * My + dR[2] * Mz;
                    const double dY = dR[3] * Mx + dR[4] * My + dR[5] * Mz;
                    const double dZ = dR[6] * Mx + dR[7] * My + dR[8] * Mz;
                    ju[j] = tu * (dX - x[h] * dZ);
                    jv[j] = tv * (dY - y[h] * dZ);
                }
                const double wu = -tu * x[h], wv = -tv * y[h];
                // packed upper triangle, row a: index q(a, b) = a * 6 - a * (a - 1) / 2 + (b - a)
                int q = 0;
#pragma unroll
                for (int a = 0; a < 3; a++) {
#pragma unroll
                    for (int b2 = a; b2 < 3; b2++) { acc[q] = fma(ju[a], ju[b2], fma(jv[a], jv[b2], acc[q])); q++; }
                    acc[q] = fma(ju[a], tu, acc[q]); q++;                        // (a, 3)
                    acc[q] = fma(jv[a], tv, acc[q]); q++;                        // (a, 4)
                    acc[q] = fma(ju[a], wu, fma(jv[a], wv, acc[q])); q++;        // (a, 5)
                    acc[21 + a] = fma(ju[a], eu[h], fma(jv[a], ev[h], acc[21 + a]));
                }
                acc[15] = fma(tu, tu, acc[15]);                                  // (3, 3);  (3, 4) = acc[16] stays 0
                acc[17] = fma(tu, wu, acc[17]);                                  // (3, 5)
                acc[18] = fma(tv, tv, acc[18]);                                  // (4, 4)
                acc[19] = fma(tv, wv, acc[19]);                                  // (4, 5)
                acc[20] = fma(wu, wu, fma(wv, wv, acc[20]));                     // (5, 5)
                acc[24] = fma(tu, eu[h], acc[24]);
                acc[25] = fma(tv, ev[h], acc[25]);
                acc[26] = fma(wu, eu[h], fma(wv, ev[h], acc[26]));
            }
        }
    }
    e2 = wave_allsum(e2);
    if (WITH_J) {
        wave_allsum32(acc, s_red);
#pragma unroll
        for (int i = 0; i < 21; i++) JtJ[i] = s_red[i];
#pragma unroll
        for (int i = 0; i < 6; i++) JtE[i] = s_red[21 + i];
    }
    return sqrt(e2);
}

__constant__ double c_pow10[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0,
                                   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};

DM_INLINE void lm_step(const double JtJ[21], const double JtE[6], int lambdaLg10, const double prev[6], double param[6]) {
    // CvLevMarq: exp(lambdaLg10 * log(10)), lambdaLg10 in [-16, 16] (the same in every lane: a scalar load)
    const double lambda = c_pow10[__builtin_amdgcn_readfirstlane(lambdaLg10) + 16];
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
// CvLevMarq's sequence per iteration is: solve -> residual at the trial pose -> (accepted) Jacobian at the same pose.  Whether an accepted
// trial ends the iteration (20 iterations or a relative parameter change below FLT_EPSILON) is known before the residual is: unless it
// does, residual and normal equations of the trial are computed in one pass and simply dropped if the trial is rejected.
// K6_LM_STATS (an instrumented build, `make lmstats`; scripts/micro/k6_lm_accept_hist.py): accepted and rejected trial steps are counted and leave the kernel in
// the upper bits of steps_done -- VERDICT r5 item 5: how often does CvLevMarq reject a step (a rejected step is one more dependent solve + residual pass)?
#ifdef K6_LM_STATS
#define K6_STAT(x) x
#else
#define K6_STAT(x)
#endif
DM_INLINE void lm_pnp(int n, const float* s_X, const float* s_uv, double* s_red, const dm::Cam& K, RodCache& rc, double pose[6], int* lm_stats = nullptr) {
    double param[6], prev[6], JtJ[21], JtE[6], JtJn[21], JtEn[6];
#pragma unroll
    for (int i = 0; i < 6; i++) param[i] = pose[i];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = 1.7976931348623157e308, errNorm = 0;
    double e_at_param = lm_eval<true>(n, s_X, s_uv, s_red, K, rc, param, JtJ, JtE);
    bool done = false;
    for (int guard = 0; guard < 64 && !done; guard++) {
#pragma unroll
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        if (iters == 0) prevErrNorm = e_at_param;
        lm_step(JtJ, JtE, lambdaLg10, prev, param);
        for (int inner = 0; inner < 40; inner++) {
            double num = 0, den = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) { num += (param[i] - prev[i]) * (param[i] - prev[i]); den += prev[i] * prev[i]; }
            // CvLevMarq: |param - prev| / (|prev| + DBL_EPSILON) < FLT_EPSILON, without the first square root and the division
            const double lim = 1.1920928955078125e-07 * (sqrt(den) + 2.220446049250313e-16);
            const bool last = (iters + 1 >= 20) || (num < lim * lim);
            if (last) errNorm = lm_eval<false>(n, s_X, s_uv, s_red, K, rc, param, nullptr, nullptr);
            else errNorm = lm_eval<true>(n, s_X, s_uv, s_red, K, rc, param, JtJn, JtEn);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { K6_STAT(lm_stats[1]++;) lm_step(JtJ, JtE, lambdaLg10, prev, param); continue; }
            }
            lambdaLg10 = max(lambdaLg10 - 1, -16);
            iters++;
            K6_STAT(lm_stats[0]++;)
            if (last) { done = true; break; }
            prevErrNorm = errNorm;
            e_at_param = errNorm;
#pragma unroll
            for (int i = 0; i < 21; i++) JtJ[i] = JtJn[i];
#pragma unroll
            for (int i = 0; i < 6; i++) JtE[i] = JtEn[i];
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
    // branch-free: every load is issued before the first result is needed (clamped addresses instead of guarded loads)
    int q[WALK_AHEAD];
#pragma unroll
    for (int s = 0; s < WALK_AHEAD; s++) q[s] = pidx[min(base + s * 64 + lane, P - 1)];
#pragma unroll
    for (int s = 0; s < WALK_AHEAD; s++) {
        const int p = min(max(q[s], 0), P - 1);
        c.p[s] = (base + s * 64 + lane < P) ? p : -1;
        c.X[s] = F.xyz[(size_t)p * 3];
        c.Y[s] = F.xyz[(size_t)p * 3 + 1];
        c.Z[s] = F.xyz[(size_t)p * 3 + 2];
        q[s] = p;
    }
    if (F.uv) {
#pragma unroll
        for (int s = 0; s < WALK_AHEAD; s++) { c.u[s] = F.uv[(size_t)q[s] * 2]; c.v[s] = F.uv[(size_t)q[s] * 2 + 1]; }
    }
}
// pixel position of a walked cell: the sampled position when the frame has one, else the cell's grid position
DM_INLINE void walk_cell_uv(const FrameDev& F, const WalkCells& c, int s, float& u, float& v) {
    if (F.uv) { u = c.u[s]; v = c.v[s]; return; }
    const int p = max(c.p[s], 0), y = p / F.W;
    u = (float)(p - y * F.W);
    v = (float)y;
}

// S waves per problem (round 6).  Wave 0 is the problem's wave as before: it walks the first 256 cells of a step's permutation and runs
// the whole LM chain.  When those 256 cells did not give max_inl inliers the walk goes on in rounds of S x 256 cells, wave w taking the w-th 256 of a round:
// fp64 residuals, threshold and ballots as in the serial walk; the waves' inlier counts meet in LDS, every wave knows the number of inliers in front of its
// own cells and compacts into those slots -- the list is the one the serial walk produces (the first max_inl inliers in permutation order), so inlier maps,
// step counts and refined poses are bit-identical.  A wave that walks a whole 640 x 480 map alone spends 2.3 ms of a 2.4 ms refinement in the walk
// (profiles/r05_k6_phases.txt: the DSAC variant refines EVERY hypothesis, 128 one-wave problems occupy an eighth of the chip); with S = 8 the same launch
// fills it.  The helper waves of a problem whose first 256 cells suffice (the soft-argmax pose on an ordinary frame) only meet wave 0 at one barrier per step.
template <int S>
__global__ __launch_bounds__(64 * S) void k_refine(int B, const int32_t* __restrict__ n_live, int live_base, int live_mul,
                                               const double* __restrict__ init_poses, const int32_t* __restrict__ perm, int steps, int max_inl,
                                               int min_inl, float thr, const int32_t* __restrict__ pert_px_c, const float* __restrict__ pert_value,
                                               FrameDev F, double* __restrict__ out_poses, int32_t* __restrict__ inlier_map,
                                               int32_t* __restrict__ steps_done, int map_stride, int group, int per_frame,
                                               const double* __restrict__ loss_gt, double* __restrict__ loss_out4,
                                               const int32_t* __restrict__ frame_of_group) {
    const int b = blockIdx.x;
    if (b >= B) return;
    // a latency chain on one wave: whatever shares its SIMD (K5 beside the replicas in dsac_backward_path1, K1 / K2 beside a deferred tail) issues after it
    __builtin_amdgcn_s_setprio(3);
    if (frame_of_group) {  // DSAC variant on a frame batch: replica list m = b / group belongs to hypothesis m, which lives in frame frame_of_group[m]
        const int f = clamp_frame(frame_of_group[group > 0 ? b / group : 0], F);
        F.xyz += (long long)f * F.xyz_stride;
        if (F.uv) F.uv += (long long)f * F.uv_stride;
    } else if (per_frame > 0) {  // frame batch: one wave per (frame, problem) -- the per-image refinement of test_ransac_softam.cpp:97-157 for F images at once
        const int f = b / per_frame;
        F.xyz += (long long)f * F.xyz_stride;
        if (F.uv) F.uv += (long long)f * F.uv_stride;
    }
    // replica lists shorter than the launch: `group` replicas per list (0: one list), list m holds live_base + live_mul * n_live[m]
    if (n_live) {
        const int m = group > 0 ? b / group : 0, local = group > 0 ? b - m * group : b;
        if (local >= live_base + live_mul * n_live[m]) return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = S > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    __shared__ float s_X[RF_MAX_INL * 3];
    __shared__ float s_uv[RF_MAX_INL * 2];
    __shared__ double s_red[32];
    __shared__ double s_pose[2][6];  // S > 1: the pose of the step, from wave 0 (by step parity: a helper reads a step's values while wave 0 may already write the next)
    __shared__ int s_head[2];        // S > 1: inliers among the step's first 256 cells (wave 0)
    // S > 1: wave 0's word to the helpers, monotonic: 4 (step + 1) + code, code 0 = this step's first 256 cells sufficed (nothing to do), 1 = walk on together,
    // 2 = the problem is over.  The helpers POLL it (s_sleep between reads): wave 0 -- the latency chain of an image -- meets no barrier on a step that
    // needs no help (a barrier per step cost the good-pose refinement 2.5 us of 83)
    __shared__ int s_word;
    __shared__ int s_wcnt[2][S > 1 ? S : 1];  // S > 1: the waves' inlier counts of a round (two sets: a round's counts are read while the next's are written)
    const dm::Cam K = make_cam_r(F);
    double pose[6];
#pragma unroll
    for (int i = 0; i < 6; i++) pose[i] = init_poses[(size_t)b * 6 + i];
    const int ppx = pert_px_c ? pert_px_c[2 * b] : -1;
    const int pch = pert_px_c ? pert_px_c[2 * b + 1] : 0;
    const float pval = pert_value ? pert_value[b] : 0.f;
    const int P = F.P;
    int done = 0;
#ifdef K6_LM_STATS
    int lm_stats[2] = {0, 0};
#endif
    RodCache rc;
    WalkCells cells;
    if (S > 1) {
        if (threadIdx.x == 0) s_word = 0;
        __syncthreads();
    }
    if (steps > 0 && wave == 0) load_walk_cells(perm, 0, lane, F, cells);
    for (int step = 0; step < steps; step++) {
        const int32_t* pidx = perm + (size_t)step * P;
        int cnt = 0;
        if (wave == 0) {
            rod_at(rc, pose);
            const double* R = rc.R;
            // wave 0 walks serially: all of the permutation when it is alone (S == 1), the first 256 cells otherwise
            for (int base = 0; base < (S == 1 ? P : min(P, 64 * WALK_AHEAD)) && cnt < max_inl; base += 64 * WALK_AHEAD) {
                if (base > 0) load_walk_cells(pidx, base, lane, F, cells);
                // the residuals of the four sub-batches are independent of the running count: all four first (their divisions and
                // square roots interleave), then the in-order compaction
                float e[WALK_AHEAD], pu[WALK_AHEAD], pv[WALK_AHEAD];
#pragma unroll
                for (int s = 0; s < WALK_AHEAD; s++) {
                    if (ppx >= 0 && cells.p[s] == ppx) { if (pch == 0) cells.X[s] = pval; else if (pch == 1) cells.Y[s] = pval; else cells.Z[s] = pval; }
                    walk_cell_uv(F, cells, s, pu[s], pv[s]);
                    e[s] = dm::residual_f(R, pose + 3, K, cells.X[s], cells.Y[s], cells.Z[s], pu[s], pv[s], 100.0);
                }
#pragma unroll
                for (int s = 0; s < WALK_AHEAD; s++) {
                    if (cnt >= max_inl) break;  // uniform: the walk stops at max_inl taken cells (core/cnn_softam.h:1121-1135)
                    const int p = cells.p[s];
                    const bool inl = (p >= 0) && (e[s] < thr);
                    const unsigned long long m = __ballot(inl);
                    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
                    const bool take = inl && (cnt + prefix < max_inl);
                    if (take) {
                        const int slot = cnt + prefix;
                        s_X[slot * 3] = cells.X[s]; s_X[slot * 3 + 1] = cells.Y[s]; s_X[slot * 3 + 2] = cells.Z[s];
                        s_uv[slot * 2] = pu[s]; s_uv[slot * 2 + 1] = pv[s];
                        if (inlier_map && (map_stride > 0 || b == 0)) atomicAdd(&inlier_map[(size_t)b * map_stride + p], 1);
                    }
                    cnt += __popcll(m);
                }
            }
            if (S > 1) {
                const bool help = cnt < max_inl && 64 * WALK_AHEAD < P;
                if (lane == 0) {
                    s_head[step & 1] = cnt;
#pragma unroll
                    for (int i = 0; i < 6; i++) s_pose[step & 1][i] = pose[i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&s_word, 4 * (step + 1) + (help ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (S > 1) {
            bool help;
            if (wave > 0) {
                int word;
                do {
                    word = __hip_atomic_load(&s_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (word < 4 * (step + 1)) __builtin_amdgcn_s_sleep(8);
                } while (word < 4 * (step + 1));
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (word >= 4 * (steps + 2)) break;    // the problem is over
                if (word >= 4 * (step + 2)) continue;  // wave 0 is already past this step: it needed no help (wave 0 cannot pass a step that does -- it waits in the round's barrier)
                help = (word & 3) == 1;
                if (!help) continue;
                cnt = s_head[step & 1];
            } else {
                help = cnt < max_inl && 64 * WALK_AHEAD < P;
            }
            if (help) {
                // the walk goes on in rounds of S x 256 cells.  A wave first COUNTS the inliers of its 256 cells; once the counts of the round have met in LDS
                // it knows how many inliers lie in front of its cells and compacts into those slots (a second pass over four ballots, no second residual)
                double hp[6];
                if (wave > 0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) hp[i] = s_pose[step & 1][i];
                    rod_at(rc, hp);
                } else {
#pragma unroll
                    for (int i = 0; i < 6; i++) hp[i] = pose[i];
                }
                unsigned long long masks[WALK_AHEAD];
                int set = 0;
                for (int base = 64 * WALK_AHEAD; base < P && cnt < max_inl; base += 64 * WALK_AHEAD * S, set ^= 1) {
                    const int mine = base + 64 * WALK_AHEAD * wave;
                    int found = 0;
                    float pu[WALK_AHEAD], pv[WALK_AHEAD];
                    if (mine < P) {
                        load_walk_cells(pidx, mine, lane, F, cells);
                        float e[WALK_AHEAD];
#pragma unroll
                        for (int s = 0; s < WALK_AHEAD; s++) {
                            if (ppx >= 0 && cells.p[s] == ppx) { if (pch == 0) cells.X[s] = pval; else if (pch == 1) cells.Y[s] = pval; else cells.Z[s] = pval; }
                            walk_cell_uv(F, cells, s, pu[s], pv[s]);
                            e[s] = dm::residual_f(rc.R, hp + 3, K, cells.X[s], cells.Y[s], cells.Z[s], pu[s], pv[s], 100.0);
                        }
#pragma unroll
                        for (int s = 0; s < WALK_AHEAD; s++) {
                            masks[s] = __ballot((cells.p[s] >= 0) && (e[s] < thr));
                            found += __popcll(masks[s]);
                        }
                    }
                    if (lane == 0) s_wcnt[set][wave] = found;
                    __syncthreads();
                    int before = cnt, total = 0;
#pragma unroll
                    for (int w = 0; w < S; w++) { const int v = s_wcnt[set][w]; if (w < wave) before += v; total += v; }
                    if (mine < P && found > 0 && before < max_inl) {
#pragma unroll
                        for (int s = 0; s < WALK_AHEAD; s++) {
                            const unsigned long long m = masks[s];
                            const bool inl = (m >> lane) & 1ull;
                            const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
                            if (inl && slot < max_inl) {
                                s_X[slot * 3] = cells.X[s]; s_X[slot * 3 + 1] = cells.Y[s]; s_X[slot * 3 + 2] = cells.Z[s];
                                s_uv[slot * 2] = pu[s]; s_uv[slot * 2 + 1] = pv[s];
                                if (inlier_map && (map_stride > 0 || b == 0)) atomicAdd(&inlier_map[(size_t)b * map_stride + cells.p[s]], 1);
                            }
                            before += __popcll(m);
                        }
                    }
                    cnt += total;
                }
                __syncthreads();  // the collected correspondences of all waves are in LDS before wave 0 reads them
            }
            if (wave > 0) continue;  // the helpers wait for wave 0's word of the next step
        }
        // the head of the next step's walk does not depend on the pose: its loads fly under the LM solve
        if (step + 1 < steps) load_walk_cells(pidx + P, 0, lane, F, cells);
        wave_sync_lds();
        const int n = min(cnt, max_inl);
        if (n < min_inl) break;  // abort for stability: too few inliers (core/cnn_softam.h:700, 1136)
        double upd[6];
#pragma unroll
        for (int i = 0; i < 6; i++) upd[i] = pose[i];
#ifdef K6_LM_STATS
        lm_pnp(n, s_X, s_uv, s_red, K, rc, upd, lm_stats);
#else
        lm_pnp(n, s_X, s_uv, s_red, K, rc, upd);
#endif
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 6; i++) nan = nan || (upd[i] != upd[i]);
        if (nan) break;
#pragma unroll
        for (int i = 0; i < 6; i++) pose[i] = upd[i];
        done++;
        __builtin_amdgcn_wave_barrier();
    }
    if (S > 1) {
        if (wave > 0) return;
        // the helpers poll for the word of a step that will not come (the problem ended early) or have left already (all steps done): tell them
        if (lane == 0) __hip_atomic_store(&s_word, 4 * (steps + 2) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) out_poses[(size_t)b * 6 + i] = pose[i];
#ifdef K6_LM_STATS
        if (steps_done) steps_done[b] = done | (min(lm_stats[0], 4095) << 8) | (min(lm_stats[1], 2047) << 20);  // accepted, rejected trial steps of the whole refinement
#else
        if (steps_done) steps_done[b] = done;
#endif
        // processImage's last stage (core/cnn_softam.h:1160-1179): maxLoss of the refined pose against this problem's ground truth, by the lane that
        // holds it -- K7's arithmetic (loss_math.h) without K7's launch behind a 90 us chain
        if (loss_out4) {
            double R1[9], t1[3], gt[6];
            max_loss_forward(pose, loss_gt + (size_t)b * 6, R1, t1, gt, loss_out4 + (size_t)b * 4);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Many problems with LONG walks (round 6): a refinement step as TWO launches, k_refine_walk (a scan of the step's permuted cells in chunks, G problems per wave;
// described at the kernel) and k_refine_lm (one wave per problem: the first max_inl inliers in chunk order, then the same lm_pnp), after one k_refine_permute per
// call.  The fused kernel above carries the LM chain's 370-380 registers on every wave (four waves per problem at most) and chases permutation entry ->
// scattered coordinate for every problem separately.  Same arithmetic, same lists: poses, step counts and inlier maps equal the fused kernel's bit for bit
// (tests/test_gpu_refine.py).  Taken for >= 32 problems on maps of >= 16 384 cells (refine_split_applies).
// state per problem: pose = out_poses[b] (running), alive[b] (1 while the loop of core/cnn_softam.h:1104-1154 would go on), steps_done[b]; per problem and
// chunk: a word (step tag | inlier count) and a list of <= max_inl records {X, Y, Z, cell, u, v}
// --------------------------------------------------------------------------------------------------
// The cells of every step's permutation laid out in walking order, once per call: cell record {X, Y, Z, bits(cell index)} + pixel position.  All problems of a
// frame walk the same permutations (core/cnn_softam.h:1108-1118 shuffles once per step), so the dependent pair of loads permutation entry -> scattered 12-byte
// coordinate (307 200 scattered reads per problem and step, the latency each round of the serial walk waits for) becomes one coalesced 16 + 8 byte read.  Every
// step's array is padded to whole rounds of 256 cells with NaN coordinates: such a cell is never "certainly an outlier" and never an inlier of the fp64 residual.
constexpr int SCAN_BOUND_SLOTS = 64, SCAN_BOUND_STRIDE = 32;  // the frame's bounds are met in 64 slots on separate cache lines (one address serialises its atomics)
__global__ __launch_bounds__(256) void k_refine_permute(int steps, int P256, FrameDev F, const int32_t* __restrict__ perm, float4* __restrict__ cells, float2* __restrict__ uvs,
                                                        int32_t* __restrict__ bounds) {
    const int i = blockIdx.x * 256 + threadIdx.x, step = blockIdx.y, f = blockIdx.z;
    const int P = F.P;
    const size_t o = ((size_t)f * steps + step) * P256 + i;
    float m = 0.f, dd = 0.f;
    if (i < P) {
        const int p = min(max(perm[(size_t)step * P + i], 0), P - 1);
        const float* xyz = F.xyz + (long long)f * F.xyz_stride;
        float2 w;
        if (F.uv) {
            const float* uv = F.uv + (long long)f * F.uv_stride;
            w = make_float2(uv[(size_t)p * 2], uv[(size_t)p * 2 + 1]);
        } else {
            const int y = p / F.W;
            w = make_float2((float)(p - y * F.W), (float)y);
        }
        const float X = xyz[(size_t)p * 3], Y = xyz[(size_t)p * 3 + 1], Z = xyz[(size_t)p * 3 + 2];
        cells[o] = make_float4(X, Y, Z, __int_as_float(p));
        uvs[o] = w;
        if (step == 0) {  // the bounds over the frame's cells themselves (cell i, not the permuted one): whatever the index lists hold is covered
            m = fabsf(xyz[(size_t)i * 3]) + fabsf(xyz[(size_t)i * 3 + 1]) + fabsf(xyz[(size_t)i * 3 + 2]);
            if (F.uv) {
                const float* uv = F.uv + (long long)f * F.uv_stride;
                dd = fabsf(uv[(size_t)i * 2] - F.cx) + fabsf(uv[(size_t)i * 2 + 1] - F.cy);
            } else {
                const int y = i / F.W;
                dd = fabsf((float)(i - y * F.W) - F.cx) + fabsf((float)y - F.cy);
            }
        }
    } else {
        const float nan = __builtin_nanf("");
        cells[o] = make_float4(nan, nan, nan, __int_as_float(0));
        uvs[o] = make_float2(0.f, 0.f);
    }
    // the frame's bounds for the scan's outlier test: max (|X| + |Y| + |Z|) and max (|u - cx| + |v - cy|) over its cells, as the integer order of non-negative
    // floats (a NaN orders above every number and switches the test off: every cell is then decided in fp64)
    if (step == 0) {
        int mi = __float_as_int(m) & 0x7fffffff, di = __float_as_int(dd) & 0x7fffffff;
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) { mi = max(mi, __shfl_xor(mi, o2)); di = max(di, __shfl_xor(di, o2)); }
        if ((threadIdx.x & 63) == 0) {
            int32_t* slot = bounds + ((size_t)f * SCAN_BOUND_SLOTS + (blockIdx.x * 4 + (threadIdx.x >> 6)) % SCAN_BOUND_SLOTS) * SCAN_BOUND_STRIDE;
            atomicMax(slot, mi);
            atomicMax(slot + 1, di);
        }
    }
}

struct WalkRound {
    float4 c[WALK_AHEAD];
    float2 w[WALK_AHEAD];
};
DM_INLINE void load_walk_round(const float4* __restrict__ cells, const float2* __restrict__ uvs, int base, int lane, WalkRound& r) {
#pragma unroll
    for (int s = 0; s < WALK_AHEAD; s++) {
        r.c[s] = cells[base + s * 64 + lane];
        r.w[s] = uvs[base + s * 64 + lane];
    }
}

// A problem's pose as the scan reads it: R | t in double (the fp64 residual's operands, rodrigues_R of the running pose) and, for the outlier test, the rows
// fx R0 | fx t0, fy R1 | fy t1, R2 | t2 rounded to float with max |t|; written by the thread that sets the pose (k_refine_split_init, lane 0 of k_refine_lm).
constexpr int SCAN_REC = 16;
DM_INLINE void write_scan_record(const double pose[6], const double R[9], const dm::Cam& K, double* __restrict__ Rt, float* __restrict__ rec) {
#pragma unroll
    for (int i = 0; i < 9; i++) Rt[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; i++) Rt[9 + i] = pose[3 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        rec[i] = (float)(K.fx * R[i]);
        rec[4 + i] = (float)(K.fy * R[3 + i]);
        rec[8 + i] = (float)R[6 + i];
    }
    rec[3] = (float)(K.fx * pose[3]);
    rec[7] = (float)(K.fy * pose[4]);
    rec[11] = (float)pose[5];
    rec[12] = fmaxf(fabsf((float)pose[3]), fmaxf(fabsf((float)pose[4]), fabsf((float)pose[5]))) * 1.0001f + 1e-30f;
    rec[13] = rec[14] = rec[15] = 0.f;
}

constexpr int SCAN_WAVES = 4;           // waves per workgroup, each with its own chunk of the permutation
constexpr int SCAN_CNT_BITS = 12;       // count field of a chunk word (counts are capped at max_inl <= RF_MAX_INL); the bits above hold the step tag
constexpr int SCAN_MAX_CHUNKS = 2048;   // prefix table of k_refine_lm (LDS)
typedef float sf2 __attribute__((ext_vector_type(2)));
typedef float sf4 __attribute__((ext_vector_type(4)));
DM_INLINE sf2 sfma(sf2 a, sf2 b, sf2 c) { return __builtin_elementwise_fma(a, b, c); }
DM_INLINE sf2 ssplat(float a) { return sf2{a, a}; }

// "These two cells are certainly NOT inliers" in fp32 (two cells per lane: v_pk_fma_f32), with a proven distance from the reference's decision (dm::residual_f:
// projection in double, float difference, double norm, compared in float).  eps = 2^-24.  A >= |error of each fp32 camera-frame coordinate|: the record's rounding
// (eps (|X| + |Y| + |Z| + |t|), times the focal length in the rows that carry it) and the three roundings of the fma chain (each <= eps x the same sum) are 4 eps M;
// A = 16 eps M with M = the frame's largest |X| + |Y| + |Z| + the pose's largest |t|.  With |zc| >= 16 A:  |x~/z~ - X/Z| <= A (1 + |x~/z~|) / (|z~| - A) <=
// T (1 + |q|), T = 2 A |1/z~| <= 1/8, so the projection is off by at most T (fx + fy + |u - cx| + |v - cy|) over both axes; v_rcp_f32's 1 ulp, the roundings of the
// products, of pu - cx and of the fused differences and the reference's float rounding of u are below rho (|u| + |cx|), rho = 2^-20; the norm adds 4 eps d.  With
// |u - cx| <= d + |pu - cx| and D = the frame's largest |pu - cx| + |pv - cy| the reference's distance is at least
// d (1 - kappa) - T (fx + fy + D) - rho (D + 2 (|cx| + |cy|)), kappa = 2 T + 2^-18 < 1/2, and the cell is no inlier if
//   d >= (thr' + T C1) (1 + 2 kappa),   thr' = thr (1 + 1e-6) + 1.01 rho C2,  C1 = fx + fy + D,  C2 = D + 2 (|cx| + |cy|)
// (squares compared, with another 2e-6 of slack).  Everything else -- inliers, cells near the threshold, points near the camera plane, NaN -- is "not certain" and
// the sub-batch is decided by residual_f.  Poses that walk the whole map have a few inliers among 300 000 cells: ~2 % of their sub-batches take the fp64 path.
// Returns the ballots of the lanes that are NOT certain, for the first and the second cell.
DM_INLINE void uncertain2(const sf4 r0, const sf4 r1, const sf4 r2, float A2, float A8, float thr1, float C1, sf2 X, sf2 Y, sf2 Z, sf2 pu_c, sf2 pv_c,
                          unsigned long long& m0, unsigned long long& m1) {
    const sf2 xs = sfma(ssplat(r0.x), X, sfma(ssplat(r0.y), Y, sfma(ssplat(r0.z), Z, ssplat(r0.w))));
    const sf2 ys = sfma(ssplat(r1.x), X, sfma(ssplat(r1.y), Y, sfma(ssplat(r1.z), Z, ssplat(r1.w))));
    const sf2 zc = sfma(ssplat(r2.x), X, sfma(ssplat(r2.y), Y, sfma(ssplat(r2.z), Z, ssplat(r2.w))));
    sf2 iz;
    iz.x = __builtin_amdgcn_rcpf(zc.x);
    iz.y = __builtin_amdgcn_rcpf(zc.y);
    const sf2 dx = sfma(-xs, iz, pu_c), dy = sfma(-ys, iz, pv_c);
    const sf2 d2 = sfma(dx, dx, dy * dy);
    sf2 aiz;
    aiz.x = fabsf(iz.x);
    aiz.y = fabsf(iz.y);
    const sf2 T = ssplat(A2) * aiz;
    const sf2 rhs = sfma(T, ssplat(C1), ssplat(thr1)) * sfma(T, ssplat(4.0f), ssplat(1.0000096f));  // 1 + 2 kappa = 1 + 4 T + 2^-17, + 2e-6
    const sf2 rhs2 = rhs * rhs;
    m0 = __ballot(!(fabsf(zc.x) >= A8) || !(d2.x > rhs2.x));
    m1 = __ballot(!(fabsf(zc.y) >= A8) || !(d2.y > rhs2.y));
}

// The walk as a SCAN: the step's permuted cells (k_refine_permute) are cut into chunks of `chunk_cells`; a wave takes one chunk and G problems of the same frame
// at once -- one coalesced read of the cells serves G poses -- and writes, per problem, the chunk's inliers in permutation order (at most max_inl) with their
// count.  k_refine_lm then takes the first max_inl inliers in chunk order: the list the serial walk of core/cnn_softam.h:1108-1134 collects.  A chunk is skipped
// for a problem when the chunks in front of it that have ALREADY finished hold max_inl inliers (a sufficient condition: results never depend on it, only the
// work does -- problems on a good pose stop after their first chunks).  Grid: x = problem group, y = 4 chunks: workgroups that read the same cells are
// dispatched together (L2).
template <int G>
__global__ __launch_bounds__(64 * SCAN_WAVES) void k_refine_walk(int B, const double* __restrict__ Rt, const float* __restrict__ rec, const int32_t* __restrict__ alive,
                                                                 const float4* __restrict__ cells_step, const float2* __restrict__ uvs_step, long long frame_stride,
                                                                 int chunk_cells, int nchunks, int P256, int tag, int max_inl, float thr, FrameDev F, int per_frame,
                                                                 int32_t* __restrict__ cnt, float* __restrict__ lists, const int32_t* __restrict__ bounds, int exact_only,
                                                                 int no_skip) {
    // workgroups go to the 8 XCDs round robin by their linear index: with the group in x alone every XCD would keep the same groups for the whole launch (and
    // groups differ: some stop after their first chunks, some scan the whole map) -- the group is rotated by the chunk row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b0 = (int)((blockIdx.x + blockIdx.y) % gridDim.x) * G;
    const int kc = (int)blockIdx.y * SCAN_WAVES + wave;
    const int f = per_frame > 0 ? b0 / per_frame : 0;
    cells_step += (long long)f * frame_stride;
    uvs_step += (long long)f * frame_stride;
    const int cbase = min(kc, nchunks - 1) * chunk_cells, cend = min(cbase + chunk_cells, P256);
    WalkRound cur, nxt;
    load_walk_round(cells_step, uvs_step, cbase, lane, cur);  // the first round's cells are on their way while the records reach LDS
    __shared__ __attribute__((aligned(16))) float s_rec[G][SCAN_REC];
    if (threadIdx.x < G * SCAN_REC) s_rec[threadIdx.x / SCAN_REC][threadIdx.x % SCAN_REC] = rec[(size_t)b0 * SCAN_REC + threadIdx.x];  // records are padded to whole groups
    __syncthreads();
    if (kc >= nchunks) return;
    int mi = bounds[((size_t)f * SCAN_BOUND_SLOTS + lane) * SCAN_BOUND_STRIDE], di = bounds[((size_t)f * SCAN_BOUND_SLOTS + lane) * SCAN_BOUND_STRIDE + 1];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mi = max(mi, __shfl_xor(mi, o)); di = max(di, __shfl_xor(di, o)); }
    const float Mmax = __int_as_float(__builtin_amdgcn_readfirstlane(mi)), Dmax = __int_as_float(__builtin_amdgcn_readfirstlane(di));
    const float C1 = (F.fx + F.fy + Dmax) * 1.000001f;
    const float thr1 = (float)((double)thr * (1.0 + 1e-6) + 1.01 * 9.5367431640625e-07 * ((double)Dmax + 2.0 * (fabs((double)F.cx) + fabs((double)F.cy)))) * 1.0000002f;
    // per problem of the wave: bit g of `live` / `act`, the chunk's inlier count in LDS (the loop over the problems is NOT unrolled: one copy of the fp64 path)
    __shared__ int s_count[SCAN_WAVES][G];
    unsigned live = 0, act = 0;
    for (int g = 0; g < G; g++) {
        if (lane == 0) s_count[wave][g] = 0;
        if (alive[b0 + g] == 0) continue;  // alive is padded to whole groups
        live |= 1u << g;
        bool skip = false;
        if (kc > 0 && !no_skip) {  // inliers of the finished chunks in front of this one (words of this step only)
            int have = 0;
            for (int j = lane; j < kc; j += 64) {
                const int w = __hip_atomic_load(&cnt[(size_t)(b0 + g) * nchunks + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                have += ((w >> SCAN_CNT_BITS) == tag) ? (w & ((1 << SCAN_CNT_BITS) - 1)) : 0;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) have += __shfl_xor(have, o);
            skip = __builtin_amdgcn_readfirstlane(have) >= max_inl;
        }
        if (!skip) act |= 1u << g;
    }
    const dm::Cam K = make_cam_r(F);
    // thresholds at or beyond the clamp (min(d, 100) < thr) and the A/B switch take the fp64 residual for every cell
    const bool use_exact = !(thr < 99.0f) || exact_only;
    for (int base = cbase; base < cend && act != 0; base += 64 * WALK_AHEAD) {
        if (base + 64 * WALK_AHEAD < cend) load_walk_round(cells_step, uvs_step, base + 64 * WALK_AHEAD, lane, nxt);  // travels under this round's arithmetic
        sf2 X[2], Y[2], Z[2], pu_c[2], pv_c[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            X[h] = sf2{cur.c[2 * h].x, cur.c[2 * h + 1].x};
            Y[h] = sf2{cur.c[2 * h].y, cur.c[2 * h + 1].y};
            Z[h] = sf2{cur.c[2 * h].z, cur.c[2 * h + 1].z};
            pu_c[h] = sf2{cur.w[2 * h].x, cur.w[2 * h + 1].x} - ssplat(F.cx);
            pv_c[h] = sf2{cur.w[2 * h].y, cur.w[2 * h + 1].y} - ssplat(F.cy);
        }
#pragma nounroll
        for (int g = 0; g < G; g++) {
            if (!((act >> g) & 1u)) continue;
            unsigned long long m[WALK_AHEAD];
            if (use_exact) {
#pragma unroll
                for (int s = 0; s < WALK_AHEAD; s++) m[s] = ~0ull;
            } else {
                const sf4 r0 = *reinterpret_cast<const sf4*>(&s_rec[g][0]), r1 = *reinterpret_cast<const sf4*>(&s_rec[g][4]), r2 = *reinterpret_cast<const sf4*>(&s_rec[g][8]);
                const float A2 = (Mmax + s_rec[g][12]) * 1.9073486328125e-06f;  // 2 A = 32 eps M
                const float A8 = 8.0f * A2;
#pragma unroll
                for (int h = 0; h < 2; h++) uncertain2(r0, r1, r2, A2, A8, thr1, C1, X[h], Y[h], Z[h], pu_c[h], pv_c[h], m[2 * h], m[2 * h + 1]);
            }
            if ((m[0] | m[1] | m[2] | m[3]) == 0ull) continue;
            // some lane is not certainly an outlier: the reference's arithmetic for its sub-batch
            double R[9], t[3];
#pragma unroll
            for (int i = 0; i < 9; i++) R[i] = Rt[(size_t)(b0 + g) * 12 + i];
#pragma unroll
            for (int i = 0; i < 3; i++) t[i] = Rt[(size_t)(b0 + g) * 12 + 9 + i];
            float* lst = lists + ((size_t)(b0 + g) * nchunks + kc) * max_inl * 6;
            int count = s_count[wave][g];
#pragma unroll
            for (int s = 0; s < WALK_AHEAD; s++) {
                if (m[s] == 0ull) continue;
                const bool inl = dm::residual_f(R, t, K, cur.c[s].x, cur.c[s].y, cur.c[s].z, cur.w[s].x, cur.w[s].y, 100.0) < thr;
                const unsigned long long mm = __ballot(inl);
                if (mm) {
                    const int slot = count + __popcll(mm & ((1ull << lane) - 1ull));
                    if (inl && slot < max_inl) {
                        float* e = lst + (size_t)slot * 6;
                        e[0] = cur.c[s].x; e[1] = cur.c[s].y; e[2] = cur.c[s].z; e[3] = cur.c[s].w; e[4] = cur.w[s].x; e[5] = cur.w[s].y;
                    }
                    count += __popcll(mm);
                }
            }
            count = __builtin_amdgcn_readfirstlane(count);
            if (lane == 0) s_count[wave][g] = count;
            if (count >= max_inl) act &= ~(1u << g);  // later inliers of this chunk cannot be among the first max_inl
        }
        cur = nxt;
    }
    if (lane == 0) {
        for (int g = 0; g < G; g++)
            if ((live >> g) & 1u) __hip_atomic_store(&cnt[(size_t)(b0 + g) * nchunks + kc], (tag << SCAN_CNT_BITS) | min(s_count[wave][g], max_inl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The LM half of a step: the first max_inl inliers in chunk order out of the scan's lists (a prefix over the chunk counts, then every list entry finds its chunk
// by bisection), the selected cells' map counters, lm_pnp.
__global__ __launch_bounds__(64) void k_refine_lm(int B, double* __restrict__ poses, int32_t* __restrict__ alive, int32_t* __restrict__ steps_done, int max_inl, int min_inl,
                                                  FrameDev F, const int32_t* __restrict__ cnt, int nchunks, int tag, const float* __restrict__ lists,
                                                  int32_t* __restrict__ inlier_map, int map_stride, double* __restrict__ Rt, float* __restrict__ rec) {
    const int b = blockIdx.x;
    if (b >= B || !alive[b]) return;
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    __shared__ float s_X[RF_MAX_INL * 3];
    __shared__ float s_uv[RF_MAX_INL * 2];
    __shared__ double s_red[32];
    __shared__ int s_start[SCAN_MAX_CHUNKS];
    int run = 0, used = 0;
    for (int j0 = 0; j0 < nchunks && run < max_inl; j0 += 64) {
        const int j = j0 + lane;
        const int w = j < nchunks ? cnt[(size_t)b * nchunks + j] : 0;
        const int c = ((w >> SCAN_CNT_BITS) == tag) ? (w & ((1 << SCAN_CNT_BITS) - 1)) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (j < nchunks) s_start[j] = run + incl - c;
        run += __builtin_amdgcn_readlane(incl, 63);
        used = min(j0 + 64, nchunks);
    }
    wave_sync_lds();
    const int n = min(run, max_inl);
    for (int e = lane; e < n; e += 64) {
        int lo = 0, hi = used - 1;  // the last chunk whose first entry is at or before e (empty chunks share their successor's start and lose to it)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const float* r = lists + (((size_t)b * nchunks + lo) * max_inl + (e - s_start[lo])) * 6;
        s_X[e * 3] = r[0]; s_X[e * 3 + 1] = r[1]; s_X[e * 3 + 2] = r[2];
        s_uv[e * 2] = r[4]; s_uv[e * 2 + 1] = r[5];
        if (inlier_map && (map_stride > 0 || b == 0)) atomicAdd(&inlier_map[(size_t)b * map_stride + __float_as_int(r[3])], 1);
    }
    if (n < min_inl) { if (lane == 0) alive[b] = 0; return; }  // abort for stability: too few inliers (core/cnn_softam.h:700, 1136)
    wave_sync_lds();
    const dm::Cam K = make_cam_r(F);
    double upd[6];
#pragma unroll
    for (int i = 0; i < 6; i++) upd[i] = poses[(size_t)b * 6 + i];
    RodCache rc;
    rod_at(rc, upd);
    lm_pnp(n, s_X, s_uv, s_red, K, rc, upd);
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 6; i++) nan = nan || (upd[i] != upd[i]);
    rod_at(rc, upd);  // the next step's scan reads the pose as a record
    if (lane == 0) {
        if (nan) alive[b] = 0;
        else {
#pragma unroll
            for (int i = 0; i < 6; i++) poses[(size_t)b * 6 + i] = upd[i];
            steps_done[b] += 1;
            write_scan_record(upd, rc.R, K, Rt + (size_t)b * 12, rec + (size_t)b * SCAN_REC);
        }
    }
}

__global__ void k_refine_split_init(int B, int Bp, FrameDev F, const double* __restrict__ init_poses, double* __restrict__ out_poses, int32_t* __restrict__ alive,
                                    int32_t* __restrict__ steps_done, double* __restrict__ Rt, float* __restrict__ rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bp) return;
    if (b >= B) {  // padding up to a whole group of the scan
        alive[b] = 0;
        for (int i = 0; i < SCAN_REC; i++) rec[(size_t)b * SCAN_REC + i] = 0.f;
        return;
    }
    const dm::Cam K = make_cam_r(F);
    double pose[6];
    for (int i = 0; i < 6; i++) { pose[i] = init_poses[(size_t)b * 6 + i]; out_poses[(size_t)b * 6 + i] = pose[i]; }
    alive[b] = 1;
    steps_done[b] = 0;
    RodCache rc;
    rod_at(rc, pose);
    write_scan_record(pose, rc.R, K, Rt + (size_t)b * 12, rec + (size_t)b * SCAN_REC);
}

static int g_scan_tune = 0;  // A/B ("k6_scan_tune"): bits 0-7 problems per wave (1, 2, 4; 0 = by the problem count), bits 8-23 chunk cells (multiple of 256; 0 = auto), bit 24: no skip check
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// chunk of the scan: 2 048 cells (eight reads of 256), smaller on maps that would not give the chip 8 192 waves otherwise
static int scan_chunk_cells(int B, int P) {
    int c = 2048;
    if ((g_scan_tune >> 8) & 0xffff) return (g_scan_tune >> 8) & 0xffff;
    while (c > 256 && (long long)((P + c - 1) / c) * B < 8192) c >>= 1;  // one wave per chunk and problem
    return c;
}
// experiments: bits 0-7 problems per wave (1, 2, 4), bits 8-23 chunk cells, bit 24: no skip check  ("k6_scan_tune")
void refine_scan_tune(int v) { g_scan_tune = v; }
// problems per wave: one up to 256 problems (19 200 light waves on a 640 x 480 map: 128 problems 36 us per step against 44 / 60 with two / four), four from
// 512 problems of frames that hold a multiple of four (the cells then cross L2 -> L1 once per four problems: 16 frames x 128 3.20 -> 2.65 ms)
static int scan_group(int B, int per_frame) {
    if (g_scan_tune & 255) {
        const int g = g_scan_tune & 255;
        return (per_frame == 0 || per_frame % g == 0) ? g : 1;
    }
    if (B < 512) return 1;
    return (per_frame == 0 || per_frame % 4 == 0) ? 4 : (per_frame % 2 == 0 ? 2 : 1);
}
static size_t scan_bounds_bytes(int frames) { return (size_t)max(frames, 1) * SCAN_BOUND_SLOTS * SCAN_BOUND_STRIDE * 4; }
size_t refine_split_scratch_bytes(int B, int steps, int frames, int P, int max_inl) {
    const int nchunks = (P + scan_chunk_cells(B, P) - 1) / scan_chunk_cells(B, P);
    const size_t Bp = (size_t)(B + 3) / 4 * 4, P256 = (size_t)(P + 255) / 256 * 256;
    return align256(Bp * 12) + align256(Bp * (12 * 8 + SCAN_REC * 4)) + align256(Bp * nchunks * 4) + scan_bounds_bytes(frames) + align256(Bp * nchunks * (size_t)max_inl * 24) +
           (size_t)max(frames, 1) * max(steps, 1) * P256 * (16 + 8) + 512;
}
bool refine_split_applies(int B, const FrameDev& F, const int32_t* pert_px_c, const double* loss_out4, int waves_per_problem) {
    return waves_per_problem == 0 && B >= 32 && F.P >= 16384 && (F.P + 255) / 256 <= SCAN_MAX_CHUNKS && !pert_px_c && !loss_out4;
}
// scratch: refine_split_scratch_bytes(B, steps, frames, P, max_inl) of device memory (frames = B / per_frame, 1 when per_frame is 0); steps_done may be null (a
// slice of the scratch is used then)
hipError_t refine_split(hipStream_t st, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr, const FrameDev& F,
                        double* out_poses, int32_t* inlier_map, int32_t* steps_done, int map_stride, int per_frame, void* scratch, int exact_only) {
    if (B <= 0) return hipSuccess;
    if (max_inl > RF_MAX_INL || max_inl >= (1 << SCAN_CNT_BITS) || steps >= (1 << (30 - SCAN_CNT_BITS))) return hipErrorInvalidValue;
    const int frames = per_frame > 0 ? B / per_frame : 1;
    const int chunk = scan_chunk_cells(B, F.P), nchunks = (F.P + chunk - 1) / chunk;
    if (nchunks > SCAN_MAX_CHUNKS) return hipErrorInvalidValue;
    const size_t Bp = (size_t)(B + 3) / 4 * 4;
    char* p = reinterpret_cast<char*>(scratch);
    int32_t* alive = reinterpret_cast<int32_t*>(p);
    int32_t* sd = steps_done ? steps_done : alive + Bp;
    p += align256(Bp * 12);
    double* Rt = reinterpret_cast<double*>(p);
    float* rec = reinterpret_cast<float*>(p + Bp * 12 * 8);
    p += align256(Bp * (12 * 8 + SCAN_REC * 4));
    int32_t* cnt = reinterpret_cast<int32_t*>(p);
    int32_t* bounds = cnt + align256(Bp * nchunks * 4) / 4;  // [frames][64 slots][32 words], cleared with the chunk words
    p += align256(Bp * nchunks * 4) + scan_bounds_bytes(frames);
    float* lists = reinterpret_cast<float*>(p); p += align256(Bp * nchunks * (size_t)max_inl * 24);
    const int P256 = (F.P + 255) / 256 * 256;
    float4* cells = reinterpret_cast<float4*>(p); p += (size_t)frames * max(steps, 1) * P256 * 16;
    float2* uvs = reinterpret_cast<float2*>(p);
    hipLaunchKernelGGL(k_refine_split_init, dim3(((int)Bp + 63) / 64), dim3(64), 0, st, B, (int)Bp, F, init_poses, out_poses, alive, sd, Rt, rec);
    if (steps <= 0) return hipGetLastError();
    hipError_t e = hipMemsetAsync(cnt, 0, align256(Bp * nchunks * 4) + scan_bounds_bytes(frames), st);  // chunk words carry the step's tag (step + 1): none is current
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_refine_permute, dim3(P256 / 256, steps, frames), dim3(256), 0, st, steps, P256, F, perm, cells, uvs, bounds);
    const long long frame_stride = (long long)steps * P256;
    const int G = scan_group(B, per_frame);
    const dim3 grid((B + G - 1) / G, (nchunks + SCAN_WAVES - 1) / SCAN_WAVES);
    for (int step = 0; step < steps; step++) {
        const float4* cs = cells + (size_t)step * P256;
        const float2* us = uvs + (size_t)step * P256;
        const int tag = step + 1;
#define WALK_ARGS B, Rt, rec, alive, cs, us, frame_stride, chunk, nchunks, P256, tag, max_inl, thr, F, per_frame, cnt, lists, bounds, exact_only, (g_scan_tune >> 24) & 1
        if (G == 4) hipLaunchKernelGGL((k_refine_walk<4>), grid, dim3(64 * SCAN_WAVES), 0, st, WALK_ARGS);
        else if (G == 2) hipLaunchKernelGGL((k_refine_walk<2>), grid, dim3(64 * SCAN_WAVES), 0, st, WALK_ARGS);
        else hipLaunchKernelGGL((k_refine_walk<1>), grid, dim3(64 * SCAN_WAVES), 0, st, WALK_ARGS);
#undef WALK_ARGS
        hipLaunchKernelGGL(k_refine_lm, dim3(B), dim3(64), 0, st, B, out_poses, alive, sd, max_inl, min_inl, F, cnt, nchunks, tag, lists, inlier_map, map_stride, Rt, rec);
    }
    return hipGetLastError();
}

hipError_t refine(hipStream_t st, int B, const double* init_poses, const int32_t* perm, int steps, int max_inl, int min_inl, float thr,
                  const int32_t* pert_px_c, const float* pert_value, const FrameDev& F, double* out_poses, int32_t* inlier_map,
                  int32_t* steps_done, int map_stride, int per_frame, const double* loss_gt_jp6, double* loss_out4, int waves_per_problem) {
    if (B <= 0) return hipSuccess;
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    // waves per problem: as many as fill the chip's 1 024 SIMDs twice over, and no more than the map has 256-cell batches behind the first one
    // (eight waves per problem would need the 380-register LM chain in 256 registers: 512 B of scratch, the good-pose refinement 83 -> 135 us -- measured, kept
    // selectable for the record: profiles/r06_k6_walk.txt)
    // ONE problem keeps one wave: the four-wave build's LM chain is 2.6 us slower on the good-pose refinement of a single image (380 against 370 registers;
    // 84.7 -> 87.3 us, the same with a barrier or a poll for the hand-off), and that chain is the latency of an image; "k6_waves" 4 buys a hard single frame 143 -> 123 us
    int S = waves_per_problem >= 1 ? waves_per_problem : (B <= 1 ? 1 : B <= 512 ? 4 : B <= 1024 ? 2 : 1);
    while (S > 1 && (long long)64 * WALK_AHEAD * (S / 2 + 1) > (long long)F.P) S /= 2;
#define DSAC_K6(S_) hipLaunchKernelGGL((k_refine<S_>), dim3(B), dim3(64 * S_), 0, st, B, (const int32_t*)nullptr, 0, 0, init_poses, perm, steps, max_inl, min_inl, thr, pert_px_c, \
                       pert_value, F, out_poses, inlier_map, steps_done, map_stride, 0, per_frame, loss_gt_jp6, loss_out4, (const int32_t*)nullptr)
    if (S >= 8) DSAC_K6(8); else if (S >= 4) DSAC_K6(4); else if (S >= 2) DSAC_K6(2); else DSAC_K6(1);
#undef DSAC_K6
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
constexpr int PLAN_TILED_MIN_CELLS = 16384;  // larger maps take the two-launch tiled plan (40 x 40: one workgroup, 5 us)

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
                                                                 int32_t* __restrict__ obj_pixels, int32_t* __restrict__ n_obj, int px_stride) {
    __shared__ int s_cnt[PLAN_THREADS];
    const int lane = threadIdx.x;
    {   // frame f of a batch (blockIdx.x): its start pose, inlier map, coordinate map and slice of the replica arrays (12 + 6*cap replicas per frame)
        const size_t f = blockIdx.x, R = 12 + 6 * (size_t)cap;
        init_pose += 6 * f; inlier_map += f * F.P; F.xyz += (long long)f * F.xyz_stride;
        rep_poses += f * R * 6; rep_px_c += f * R * 2; rep_value += f * R; obj_pixels += f * px_stride; n_obj += f;
    }
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

// ---- the same plan for large maps: two launches over tiles of 64 columns --------------------------------------------------------
// A single workgroup scanning a 640 x 480 map column-major reads it with a 2.5 KB stride: 158 us, a quarter of that frame's training
// geometry.  Tiled: workgroup b owns columns 64 b .. 64 b + 63; lane = column, so a wave reads a row segment of 256 contiguous bytes, and
// the 16 waves split the rows.  Launch 1 counts the inliers per (column, row segment); launch 2 turns the counts into the column-major
// rank of every inlier (columns before the tile, columns before it inside the tile, row segments above it in its column, then a walk down
// its own segment) and emits the selected ones exactly as the one-workgroup form does.  scratch: PLAN_SEGS + 1 ints per column.
constexpr int PLAN_SEGS = PLAN_THREADS / 64;  // row segments = waves per workgroup

__global__ __launch_bounds__(PLAN_THREADS) void k_refine_fd_count(const int32_t* __restrict__ inlier_map, FrameDev F, int32_t* __restrict__ scratch) {
    inlier_map += (size_t)blockIdx.y * F.P;                       // map m of a batch (DSAC variant: one inlier map per hypothesis)
    scratch += (size_t)blockIdx.y * F.W * (PLAN_SEGS + 1);
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int rows = (F.H + PLAN_SEGS - 1) / PLAN_SEGS;
    const int y0 = min(F.H, seg * rows), y1 = min(F.H, y0 + rows);
    int cnt = 0;
    if (x < F.W)
        for (int y = y0; y < y1; y++) cnt += inlier_map[(size_t)y * F.W + x] != 0;
    __shared__ int s_c[PLAN_SEGS][64];
    s_c[seg][lane] = cnt;
    __syncthreads();
    if (x < F.W) {
        scratch[(size_t)F.W + (size_t)x * PLAN_SEGS + seg] = cnt;          // per (column, segment)
        if (seg == 0) {
            int tot = 0;
#pragma unroll
            for (int k = 0; k < PLAN_SEGS; k++) tot += s_c[k][lane];
            scratch[x] = tot;                                                // per column
        }
    }
}

// SET: the DSAC variant's replica list (18 set perturbations first, start poses written later by k_refine_fd_init_set)
template <bool SET>
DM_INLINE void emit_obj_replicas(int slot, int p, const FrameDev& F, const double init[6], float eps_obj, double* __restrict__ rep_poses,
                                 int32_t* __restrict__ rep_px_c, float* __restrict__ rep_value, int32_t* __restrict__ obj_pixels) {
    obj_pixels[slot] = p;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v0 = F.xyz[(size_t)p * 3 + c];
        const float vf = v0 + eps_obj;
        const float vb = vf - 2 * eps_obj;
        const int r = (SET ? 18 : 12) + slot * 6 + c * 2;
        if (!SET) {
#pragma unroll
            for (int k = 0; k < 6; k++) { rep_poses[(size_t)r * 6 + k] = init[k]; rep_poses[(size_t)(r + 1) * 6 + k] = init[k]; }
        }
        rep_px_c[2 * r] = p; rep_px_c[2 * r + 1] = c; rep_value[r] = vf;
        rep_px_c[2 * (r + 1)] = p; rep_px_c[2 * (r + 1) + 1] = c; rep_value[r + 1] = vb;
    }
}

template <bool SET>
__global__ __launch_bounds__(PLAN_THREADS) void k_refine_fd_emit(const double* __restrict__ init_pose, const int32_t* __restrict__ inlier_map, FrameDev F,
                                                                 int skip, float eps_hyp, float eps_obj, int cap, const int32_t* __restrict__ scratch,
                                                                 double* __restrict__ rep_poses, int32_t* __restrict__ rep_px_c,
                                                                 float* __restrict__ rep_value, int32_t* __restrict__ obj_pixels,
                                                                 int32_t* __restrict__ n_obj, const int32_t* __restrict__ set4, int px_stride,
                                                                 const int32_t* __restrict__ frame_of) {
    const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
    const int x0 = blockIdx.x * 64, x = x0 + lane;
    double init[6] = {0, 0, 0, 0, 0, 0};
    if (SET) {  // hypothesis m of a batch (blockIdx.y): its set, inlier map, counts and slice of the replica arrays (18 + 6*cap replicas each)
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        set4 += 4 * m; inlier_map += m * F.P; scratch += m * F.W * (PLAN_SEGS + 1); rep_px_c += m * R * 2; rep_value += m * R; obj_pixels += m * cap; n_obj += m;
        if (frame_of) F.xyz += (long long)clamp_frame(frame_of[m], F) * F.xyz_stride;  // frame batch: hypothesis m reads its own frame's coordinates
        if (blockIdx.x == 0 && tid < 18) {
            const int pt = tid / 6, c = (tid % 6) >> 1;
            const int p = min(max(set4[pt], 0), F.P - 1);
            const float vf = F.xyz[(size_t)p * 3 + c] + eps_obj;
            rep_px_c[2 * tid] = p; rep_px_c[2 * tid + 1] = c;
            rep_value[tid] = (tid & 1) ? vf - 2 * eps_obj : vf;
        }
    } else {
        // frame f of a batch (blockIdx.y): its start pose, inlier map, coordinate map, counts and slice of the replica arrays
        const size_t f = blockIdx.y, R = 12 + 6 * (size_t)cap;
        init_pose += 6 * f; inlier_map += f * F.P; F.xyz += (long long)f * F.xyz_stride; scratch += f * F.W * (PLAN_SEGS + 1);
        rep_poses += f * R * 6; rep_px_c += f * R * 2; rep_value += f * R; obj_pixels += f * (size_t)px_stride; n_obj += f;
#pragma unroll
        for (int i = 0; i < 6; i++) init[i] = init_pose[i];
    }
    if (!SET && blockIdx.x == 0 && tid < 12) {  // dRefineHyp: replica 2i = +step on parameter i, 2i+1 = (+step) - 2 step (as the one-workgroup form)
        const int i = tid >> 1;
        const double step = (i < 3) ? (double)eps_hyp : (double)(eps_hyp * 1000);
        double v = init[i] + step;
        if (tid & 1) v -= 2 * step;
#pragma unroll
        for (int k = 0; k < 6; k++) rep_poses[(size_t)tid * 6 + k] = (k == i) ? v : init[k];
        rep_px_c[2 * tid] = -1; rep_px_c[2 * tid + 1] = 0; rep_value[tid] = 0.f;
    }
    // inliers in the columns before this tile (all threads), and in all columns (for n_obj)
    __shared__ int s_part[PLAN_SEGS], s_tot[PLAN_SEGS], s_col[64];
    int before = 0, total = 0;
    for (int i = tid; i < F.W; i += PLAN_THREADS) { const int v = scratch[i]; total += v; if (i < x0) before += v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); total += __shfl_xor(total, o, 64); }
    if (lane == 0) { s_part[seg] = before; s_tot[seg] = total; }
    if (seg == 0) {  // exclusive prefix over the tile's 64 columns
        const int own = x < F.W ? scratch[x] : 0;
        int incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        s_col[lane] = incl - own;
    }
    __syncthreads();
    before = 0; total = 0;
#pragma unroll
    for (int k = 0; k < PLAN_SEGS; k++) { before += s_part[k]; total += s_tot[k]; }
    if (blockIdx.x == 0 && tid == 0) n_obj[0] = min(total / skip, cap);
    if (x >= F.W) return;
    int inCount = before + s_col[lane];
    for (int k = 0; k < seg; k++) inCount += scratch[(size_t)F.W + (size_t)x * PLAN_SEGS + k];
    const int rows = (F.H + PLAN_SEGS - 1) / PLAN_SEGS;
    const int y0 = min(F.H, seg * rows), y1 = min(F.H, y0 + rows);
    for (int y = y0; y < y1; y++) {
        const int p = y * F.W + x;
        if (inlier_map[p] == 0) continue;
        inCount++;
        if (inCount % skip != 0) continue;
        const int slot = inCount / skip - 1;
        if (slot >= cap) continue;
        emit_obj_replicas<SET>(slot, p, F, init, eps_obj, rep_poses, rep_px_c, rep_value, obj_pixels);
    }
}

size_t refine_fd_plan_scratch_ints(const FrameDev& F) { return F.P > PLAN_TILED_MIN_CELLS ? (size_t)F.W * (PLAN_SEGS + 1) : 0; }

hipError_t refine_fd_plan(hipStream_t st, const double* init_pose, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_hyp,
                          float eps_obj, int cap, double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels, int32_t* n_obj,
                          int32_t* scratch, int frames, int px_stride) {
    if (frames < 1) frames = 1;
    if (px_stride <= 0) px_stride = cap;
    if (scratch && F.P > PLAN_TILED_MIN_CELLS) {
        const int tiles = (F.W + 63) / 64;
        hipLaunchKernelGGL(k_refine_fd_count, dim3(tiles, frames), dim3(PLAN_THREADS), 0, st, inlier_map, F, scratch);
        hipLaunchKernelGGL(k_refine_fd_emit<false>, dim3(tiles, frames), dim3(PLAN_THREADS), 0, st, init_pose, inlier_map, F, skip, eps_hyp, eps_obj, cap, scratch,
                           rep_poses, rep_px_c, rep_value, obj_pixels, n_obj, (const int32_t*)nullptr, px_stride, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_refine_fd_plan, dim3(frames), dim3(PLAN_THREADS), 0, st, init_pose, inlier_map, F, skip, eps_hyp, eps_obj, cap, rep_poses, rep_px_c,
                       rep_value, obj_pixels, n_obj, px_stride);
    return hipGetLastError();
}

// launches the replicas: grid = 12 + 6*cap waves, those beyond 12 + 6*n_obj exit immediately
hipError_t refine_fd_run(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                         int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out, int frames) {
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    const int R = 12 + 6 * cap;
    if (frames > 1) {  // one replica list per frame: list m = b / R refines against frame m, replicas beyond 12 + 6 * n_obj[m] exit at once
        hipLaunchKernelGGL(k_refine<1>, dim3(R * frames), dim3(64), 0, st, R * frames, n_obj, 12, 6, rep_poses, perm, steps, max_inl, min_inl, thr, rep_px_c, rep_value, F,
                           rep_out, (int32_t*)nullptr, (int32_t*)nullptr, 0, R, R, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_refine<1>, dim3(R), dim3(64), 0, st, R, n_obj, 12, 6, rep_poses, perm, steps, max_inl, min_inl, thr, rep_px_c, rep_value, F, rep_out,
                       (int32_t*)nullptr, (int32_t*)nullptr, 0, 0, 0, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr);
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
                                                           int32_t* __restrict__ obj_pixels, int32_t* __restrict__ n_obj,
                                                           const int32_t* __restrict__ frame_of) {
    const int lane = threadIdx.x;
    const int P = F.P;
    {   // hypothesis m of a batch (blockIdx.y): its set, inlier map and slice of the replica arrays (18 + 6*cap replicas each)
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        set4 += 4 * m; inlier_map += m * P; rep_px_c += m * R * 2; rep_value += m * R; obj_pixels += m * cap; n_obj += m;
        if (frame_of) F.xyz += (long long)clamp_frame(frame_of[m], F) * F.xyz_stride;  // frame batch: hypothesis m reads its own frame's coordinates
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

// start pose of replica r: P3P (least-squares alignment, as OpenCV's; closed form since round 5) of the set read through the replica's perturbation
__global__ __launch_bounds__(64) void k_refine_fd_init_set(int cap, const int32_t* __restrict__ n_obj, const int32_t* __restrict__ set4,
                                                           const int32_t* __restrict__ rep_px_c, const float* __restrict__ rep_value, FrameDev F,
                                                           double* __restrict__ rep_poses, const int32_t* __restrict__ frame_of) {
    {
        const size_t m = blockIdx.y, R = 18 + 6 * (size_t)cap;
        set4 += 4 * m; rep_px_c += m * R * 2; rep_value += m * R; rep_poses += m * R * 6; n_obj += m;
        if (frame_of) {
            const long long f = clamp_frame(frame_of[m], F);
            F.xyz += f * F.xyz_stride;
            if (F.uv) F.uv += f * F.uv_stride;
        }
    }
    // four lanes per replica, one per quartic root (side by side, as in K5)
    const int r = blockIdx.x * 16 + (threadIdx.x >> 2), root = threadIdx.x & 3;
    const bool active = r < 18 + 6 * min(n_obj[0], cap);
    bool cand = false;
    double Rc[9], Tc[3], reproj = 0;
    if (active) {
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
        const dm::Cam K = make_cam_r(F);
        dm::P3PSetup S;
        if (dm::p3p_setup(X, uv, K, S) && root < S.n) {
            const double x = (root == 0) ? S.roots[0] : (root == 1) ? S.roots[1] : (root == 2) ? S.roots[2] : S.roots[3];
            cand = dm::p3p_eval_root<false>(S, X, uv, K, x, Rc, Tc, reproj);  // least-squares alignment in closed form (dmath.h), as in K1 / K5
        }
    }
    const int win = dm::best_root_of_quad(cand, reproj);
    if (active && (win == root || (win < 0 && root == 0))) {
        double cv6[6] = {0, 0, 0, 0, 0, 0};  // no root: safeSolvePnP's zero pose
        if (win == root) {
            dm::rodrigues_m2v(Rc, cv6);
            cv6[3] = Tc[0]; cv6[4] = Tc[1]; cv6[5] = Tc[2];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) rep_poses[(size_t)r * 6 + k] = cv6[k];
    }
}

hipError_t refine_fd_plan_set(hipStream_t st, const int32_t* set4, const int32_t* inlier_map, const FrameDev& F, int skip, float eps_obj, int cap,
                              double* rep_poses, int32_t* rep_px_c, float* rep_value, int32_t* obj_pixels, int32_t* n_obj, int M, int32_t* scratch,
                              const int32_t* frame_of) {
    if (M <= 0) return hipSuccess;
    if (scratch && F.P > PLAN_TILED_MIN_CELLS) {
        // large maps: the tiled two-launch plan of the soft-argmax path, one grid row per hypothesis (round 2 scanned each map with one workgroup, column-major
        // with a 2.5 KB stride: 158 us per 640 x 480 map)
        const int tiles = (F.W + 63) / 64;
        hipLaunchKernelGGL(k_refine_fd_count, dim3(tiles, M), dim3(PLAN_THREADS), 0, st, inlier_map, F, scratch);
        hipLaunchKernelGGL(k_refine_fd_emit<true>, dim3(tiles, M), dim3(PLAN_THREADS), 0, st, (const double*)nullptr, inlier_map, F, skip, 0.f, eps_obj, cap, scratch,
                           (double*)nullptr, rep_px_c, rep_value, obj_pixels, n_obj, set4, cap, frame_of);
    } else
    hipLaunchKernelGGL(k_refine_fd_plan_set, dim3(1, M), dim3(PLAN_THREADS), 0, st, set4, inlier_map, F, skip, eps_obj, cap, rep_px_c, rep_value, obj_pixels, n_obj, frame_of);
    const int R = 18 + 6 * cap;
    hipLaunchKernelGGL(k_refine_fd_init_set, dim3((R + 15) / 16, M), dim3(64), 0, st, cap, n_obj, set4, rep_px_c, rep_value, F, rep_poses, frame_of);
    return hipGetLastError();
}

hipError_t refine_fd_run_set(hipStream_t st, int cap, const int32_t* n_obj, const double* rep_poses, const int32_t* perm, int steps, int max_inl,
                             int min_inl, float thr, const int32_t* rep_px_c, const float* rep_value, const FrameDev& F, double* rep_out, int M,
                             const int32_t* frame_of) {
    if (max_inl > RF_MAX_INL) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    const int R = 18 + 6 * cap;
    const long long B = (long long)R * M;
    if (B > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_refine<1>, dim3((unsigned)B), dim3(64), 0, st, (int)B, n_obj, 18, 6, rep_poses, perm, steps, max_inl, min_inl, thr, rep_px_c, rep_value, F,
                       rep_out, (int32_t*)nullptr, (int32_t*)nullptr, 0, R, 0, (const double*)nullptr, (double*)nullptr, frame_of);
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
    {   // frame f of a batch (blockIdx.y): its replica results and Jacobians
        const size_t f = blockIdx.y;
        rep_out += f * (12 + 6 * (size_t)cap) * 6; n_obj += f; J_hyp += f * 36; J_obj += f * (size_t)cap * 18;
    }
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
                            double* J_hyp, double* J_obj, int frames) {
    const int pairs = 6 + 3 * cap;
    hipLaunchKernelGGL(k_refine_fd_finish, dim3((pairs + 63) / 64, frames < 1 ? 1 : frames), dim3(64), 0, st, rep_out, n_obj, cap, skip, eps_hyp, eps_obj, J_hyp, J_obj);
    return hipGetLastError();
}

}  // namespace dk

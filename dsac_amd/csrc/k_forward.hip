// k_forward.hip -- forward kernels of the gfx950 DSAC engine.
//
//   pose_prep   : fp64 Rodrigues of the N cv poses -> 48-byte float records (intrinsics folded in)
//   k_reproject : K2, the N x P reprojection / error-image / soft-inlier kernel (HBM-write bound)
//   reduce_soft : deterministic second stage of the soft-inlier sums
//   softmax     : K3, softmax + entropy + soft-argmax pose (fp64, one workgroup)
//
// K2 replaces the N getDiffMap calls of the reference (core/cnn_softam.h:1067-1069, getDiffMap :319-362).
// Data layout in HBM:  xyz  P x 3 f32 (AoS, 12 B/pixel, read once per hypothesis tile, L2-resident),
//                      err  N x P f32 hypothesis-major (written once, streaming, non-temporal),
//                      staged poses N x 12 f32.
// Work decomposition: a workgroup of 256 lanes owns a tile of 1024 consecutive pixels (4 per lane, so that
// every global access is a 16-byte dwordx4 and one wave store covers 1 KiB of one error-image row) and a
// tile of HT hypotheses whose 3x4 records are staged in LDS and broadcast to all lanes.  The block index is
// decoded XCD-aware: blocks that share a pixel tile run on the same XCD (block b -> XCD b % 8) back to
// back, so the xyz tile is fetched from HBM once and re-read from that XCD's L2.
#include "kernels.h"
#include <hip/hip_ext.h>
#include "dmath.h"

namespace dk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pose_prep(int N, const double* __restrict__ poses, float fx, float fy, float* __restrict__ staged) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    double r[3] = {poses[6 * h], poses[6 * h + 1], poses[6 * h + 2]};
    double R[9];
    dm::rodrigues_v2m<false>(r, R, nullptr);
    const double t0 = poses[6 * h + 3], t1 = poses[6 * h + 4], t2 = poses[6 * h + 5];
    float* o = staged + (size_t)h * POSE_STRIDE;
    const double dfx = fx, dfy = fy;
    o[0] = (float)(dfx * R[0]); o[1] = (float)(dfx * R[1]); o[2] = (float)(dfx * R[2]); o[3] = (float)(dfx * t0);
    o[4] = (float)(dfy * R[3]); o[5] = (float)(dfy * R[4]); o[6] = (float)(dfy * R[5]); o[7] = (float)(dfy * t1);
    o[8] = (float)R[6]; o[9] = (float)R[7]; o[10] = (float)R[8]; o[11] = (float)t2;
}

// What the float records leave behind: lo = (float)(A - (double)(float)A) for the twelve entries of k_pose_prep's record (the same fp64 arithmetic, so
// the pair is consistent with the records K1 stages itself).  k2_flags bit 27.
__global__ __launch_bounds__(256) void k_pose_prep_lo(int N, const double* __restrict__ poses, float fx, float fy, float* __restrict__ staged_lo) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    double r[3] = {poses[6 * h], poses[6 * h + 1], poses[6 * h + 2]};
    double R[9];
    dm::rodrigues_v2m<false>(r, R, nullptr);
    const double dfx = fx, dfy = fy;
    const double A[12] = {dfx * R[0], dfx * R[1], dfx * R[2], dfx * poses[6 * h + 3], dfy * R[3], dfy * R[4], dfy * R[5], dfy * poses[6 * h + 4],
                          R[6], R[7], R[8], poses[6 * h + 5]};
    float* o = staged_lo + (size_t)h * POSE_STRIDE;
#pragma unroll
    for (int k = 0; k < 12; k++) o[k] = (float)(A[k] - (double)(float)A[k]);
}

hipError_t pose_prep_lo(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged_lo) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pose_prep_lo, dim3((N + 255) / 256), dim3(256), 0, st, N, poses, F.fx, F.fy, staged_lo);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// Round 6: the pose records as fp16 PIECES on fixed binary grids, for the exact-transform form of K2 (k2_flags bit 28; k_reproject_st<.., EX>).
// The reference projects in double (core/cnn_softam.h:319-362).  E = R.X + t is evaluated here without any rounding that matters by writing both factors as
// sums of short fixed-point pieces whose products are exact in fp32 and by sorting the products into TWO matrix-core accumulations:
//   * the HIGH group -- every product that is a multiple of 2^(E-5) -- is summed EXACTLY by one v_mfma_f32_16x16x16_f16 per row (all addends on one grid,
//     |sum| < 2^23 grid units):  aA XA + aT XM + aBT XT + tA + tA2;
//   * the CROSS group -- what is left, |sum| of a few millimetres -- by one v_mfma_f32_16x16x32_f16 per row, scaled by 2^14 so that every half is a
//     NORMAL number (the matrix core flushes subnormal halves):  aC XA + aA' XM + aB' XT + aA XL1 + aA XL2 + aB XA' + aB XM + aB XL1 + tC + tD;
//   * E = D_hi + 2^-14 D_cross: one fused multiply-add, the single rounding to float of the camera-frame point.
// Why the split is where it is: the fp16 matrix core aligns the products of one instruction to the LARGEST of them, keeps ~24 bits and TRUNCATES the rest
// (profiles/r06_mfma_f16_numerics.txt: 2^14 + 0.75 ulp -> 2^14; 2^14 - 2^14 + 2^-12 -> 0).  The first design of this form put every product but aA XA + tA
// into the cross group (largest addends ~2^5 mm): the truncation of its small addends, whose signs follow the hypothesis, shifted a hypothesis' points by
// ~1e-6 mm systematically -- enough to move the softmax weight of two tied hypotheses by 1.5-2.2e-4 (stated 1e-4; profiles/r06_k2_diag.txt).  With the three
// largest remainders moved into the exact group the cross sum stays below ~4 mm and what its window drops is below 2^-22 mm.
// Pieces (E = 0 for the z row, `ex` = ceil(log2 f) <= 10 for the focal-length-folded x / y rows; n x 2^g = an integer |n| <= 2047 times the grid 2^g):
//   a = aA + aB + aC     aA: 2^(E-10)   aB: 2^(E-21)   aC: 2^(E-32)        aA = aT + aA', aT: 2^(E-5)        aB = aBT + aB', aBT: 2^(E-15)
//   X = XA + XM + XL1 + XL2    XA: 32 rint(X / 32) = XT + XA', XT: 1024 rint(X / 1024)    XM: rint(X - XA)    XL1: 2^-12 rint(.)    XL2: the rest (< 2^-13)
//   t = tA + tA2 + tC + tD     tA: 2^(E+6)   tA2: 2^(E-5)   tC: 2^(E-16)   tD: 2^(E-27)
// What this kernel writes is the A-operand image in the matrix core's own lane layout: for hypothesis h, row r in (x, y, z), quarter q (= coordinate X, Y, Z,
// or the translation for q = 3): eight halves for the K = 32 instruction (k = 8q .. 8q + 7) and four for the K = 16 one (k = 4q .. 4q + 3), whose B operand
// is the FIRST HALF of the other's:
//   B[q < 3]     = (XA 2^-5,   XM 2^3,    XT 2^-4,   XL1 2^10,  XL2 2^10,  XA',     XM,       XL1   )
//   hi[q < 3]    = (aA 2^5,    aT 2^-3,   aBT 2^4,   0)
//   cross[q < 3] = (aC 2^19,   aA' 2^11,  aB' 2^18,  aA 2^4,    aA 2^4,    aB 2^14, aB 2^14,  aB 2^14)
//   B[q = 3]     = (2^10, 2^3, 2^11, 0, 0, 0, 1, 0) -- the pieces of the constant 2^15 + 1 run through the same code as a coordinate
//   hi[q = 3]    = (0, tA2 2^-3, tA 2^-11, 0)          cross[q = 3] = (0, tC 2^11, 0, 0, 0, 0, tD 2^14, 0)
// The x and y rows carry the minus sign of hp_chunk's (-xc, -yc, zc).  Valid for focal lengths up to 2^10 px, |t| < 2^16 mm, |X| < 2^16 mm (the kernel
// sends chunks with larger coordinates down the fp32 path); pieces are clamped to +-2047 grid units, so out-of-range input degrades, it does not overflow.
// --------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int SPLIT_CROSS_BYTES = 12 * 16, SPLIT_HI_BYTES = 12 * 8;  // per hypothesis: 3 rows x 4 quarters
constexpr int SPLIT_MAX_EX = 10;  // pose_split_exponent above this: no exact form (api.hip falls back to the fp32 records)
DM_INLINE double split_piece(double& rem, int grid_exp) {  // the multiple of 2^grid_exp nearest to rem (at most 2047 units), which is taken out of rem
    double n = rint(ldexp(rem, -grid_exp));
    n = fmin(fmax(n, -2047.0), 2047.0);
    const double p = ldexp(n, grid_exp);
    rem -= p;
    return p;
}
DM_INLINE _Float16 hf(double v, int e) { return (_Float16)(float)ldexp(v, e); }
__global__ __launch_bounds__(256) void k_pose_prep_split(int N, const double* __restrict__ poses, float fx, float fy, int ex, h8* __restrict__ cross, h4* __restrict__ hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (hypothesis, row)
    const int h = i / 3, r = i - 3 * h;
    if (h >= N) return;
    double rv[3] = {poses[6 * h], poses[6 * h + 1], poses[6 * h + 2]};
    double R[9];
    dm::rodrigues_v2m<false>(rv, R, nullptr);
    const double f = r == 0 ? -(double)fx : r == 1 ? -(double)fy : 1.0;
    const int E = r < 2 ? ex : 0;
    const double a[4] = {f * R[3 * r], f * R[3 * r + 1], f * R[3 * r + 2], f * poses[6 * h + 3 + r]};
    h8* oc = cross + ((size_t)h * 3 + r) * 4;
    h4* oh = hi + ((size_t)h * 3 + r) * 4;
    const _Float16 z = (_Float16)0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        double rem = a[q];
        const double aA = split_piece(rem, E - 10), aB = split_piece(rem, E - 21), aC = split_piece(rem, E - 32);
        double t1 = aA, t2 = aB;
        const double aT = split_piece(t1, E - 5), aBT = split_piece(t2, E - 15);  // t1 = aA' (|aA'| <= 2^(E-6)), t2 = aB' (<= 2^(E-16))
        oh[q] = h4{hf(aA, 5), hf(aT, -3), hf(aBT, 4), z};
        oc[q] = h8{hf(aC, 19), hf(t1, 11), hf(t2, 18), hf(aA, 4), hf(aA, 4), hf(aB, 14), hf(aB, 14), hf(aB, 14)};
    }
    double rem = a[3];
    const double tA = split_piece(rem, E + 6), tA2 = split_piece(rem, E - 5), tC = split_piece(rem, E - 16), tD = split_piece(rem, E - 27);
    // against B[q = 3] = the pieces of SPLIT_T_COORD = (2^10, 2^3, 2^11, 0, 0, 0, 1, 0)
    oh[3] = h4{z, hf(tA2, -3), hf(tA, -11), z};
    oc[3] = h8{z, hf(tC, 11), z, z, z, z, hf(tD, 14), z};
}

size_t pose_split_bytes(int N) { return (size_t)N * (SPLIT_CROSS_BYTES + SPLIT_HI_BYTES); }
bool pose_split_available(const FrameDev& F) { return pose_split_exponent(F) <= SPLIT_MAX_EX; }
int pose_split_exponent(const FrameDev& F) {  // ceil(log2(max focal length)), at least 0; > SPLIT_MAX_EX = 10: the exact form is not available
    int e = 0;
    while (ldexp(1.0, e) < (double)fmaxf(F.fx, F.fy)) e++;
    return e;
}
hipError_t pose_prep_split(hipStream_t st, int N, const double* poses, const FrameDev& F, void* split) {
    if (N <= 0) return hipSuccess;
    h8* cross = reinterpret_cast<h8*>(split);
    h4* hi = reinterpret_cast<h4*>(reinterpret_cast<char*>(split) + (size_t)N * SPLIT_CROSS_BYTES);
    hipLaunchKernelGGL(k_pose_prep_split, dim3((3 * N + 255) / 256), dim3(256), 0, st, N, poses, F.fx, F.fy, pose_split_exponent(F), cross, hi);
    return hipGetLastError();
}

hipError_t pose_prep(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pose_prep, dim3((N + 255) / 256), dim3(256), 0, st, N, poses, F.fx, F.fy, staged);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K2
// --------------------------------------------------------------------------------------------------
constexpr int K2_THREADS = 256;

DM_INLINE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One residual.  r0/r1 carry the focal length, (pu, pv) are the pixel position minus the principal point.
DM_INLINE float residual(const f4 r0, const f4 r1, const f4 r2, float X, float Y, float Z, float pu, float pv, float clampv) {
    const float xc = fmaf(r0.x, X, fmaf(r0.y, Y, fmaf(r0.z, Z, r0.w)));
    const float yc = fmaf(r1.x, X, fmaf(r1.y, Y, fmaf(r1.z, Z, r1.w)));
    const float zc = fmaf(r2.x, X, fmaf(r2.y, Y, fmaf(r2.z, Z, r2.w)));
    // projectPoints: z = Z ? 1/Z : 1
    const float iz = (zc == 0.0f) ? 1.0f : __builtin_amdgcn_rcpf(zc);
    const float du = fmaf(-xc, iz, pu);
    const float dv = fmaf(-yc, iz, pv);
    const float d2 = fmaf(dv, dv, du * du);
    return fminf(__builtin_amdgcn_sqrtf(d2), clampv);
}

// sigmoid(beta * (tau - e)) = 1 / (1 + 2^(kA * e + kB))  with kA = beta*log2(e), kB = -beta*tau*log2(e)
DM_INLINE float soft_inlier(float e, float kA, float kB) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kA, e, kB)));
}

// Packed-fp32 forms: two pixels per VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  K2 is
// VALU-issue bound once the stores stream (rocprof: SQ_ACTIVE_INST_VALU ~ all SIMD cycles), and a packed op
// issues in the time of a scalar one, so the 9-FMA rigid transform and the residual arithmetic run at
// half the issue cost; only rcp / sqrt / exp2 (transcendental pipe) and min stay per pixel.
DM_INLINE f2 splat(float a) { return f2{a, a}; }
DM_INLINE f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

DM_INLINE f2 residual2(const f4 r0, const f4 r1, const f4 r2, f2 X, f2 Y, f2 Z, f2 pu, f2 pv, float clampv) {
    const f2 xc = pk_fma(splat(r0.x), X, pk_fma(splat(r0.y), Y, pk_fma(splat(r0.z), Z, splat(r0.w))));
    const f2 yc = pk_fma(splat(r1.x), X, pk_fma(splat(r1.y), Y, pk_fma(splat(r1.z), Z, splat(r1.w))));
    const f2 zc = pk_fma(splat(r2.x), X, pk_fma(splat(r2.y), Y, pk_fma(splat(r2.z), Z, splat(r2.w))));
    f2 iz;
    iz.x = (zc.x == 0.0f) ? 1.0f : __builtin_amdgcn_rcpf(zc.x);
    iz.y = (zc.y == 0.0f) ? 1.0f : __builtin_amdgcn_rcpf(zc.y);
    const f2 du = pk_fma(-xc, iz, pu);
    const f2 dv = pk_fma(-yc, iz, pv);
    const f2 d2 = pk_fma(dv, dv, du * du);
    f2 e;
    e.x = fminf(__builtin_amdgcn_sqrtf(d2.x), clampv);
    e.y = fminf(__builtin_amdgcn_sqrtf(d2.y), clampv);
    return e;
}

DM_INLINE f2 soft_inlier2(f2 e, float kA, float kB) {
    const f2 t = pk_fma(splat(kA), e, splat(kB));
    f2 ex;
    ex.x = __builtin_amdgcn_exp2f(t.x);
    ex.y = __builtin_amdgcn_exp2f(t.y);
    const f2 d = ex + splat(1.0f);
    f2 s;
    s.x = __builtin_amdgcn_rcpf(d.x);
    s.y = __builtin_amdgcn_rcpf(d.y);
    return s;
}

template <int PX, int HT, bool ERR, bool SOFT, bool UV, bool SPOSE>
__global__ __launch_bounds__(K2_THREADS) void k_reproject(const float* __restrict__ staged, const float* __restrict__ xyz,
                                                          const float* __restrict__ uv, float* __restrict__ err,
                                                          float* __restrict__ soft_part, int N, int P, int W, int PT, int NT, float cx,
                                                          float cy, float clampv, float kA, float kB, int kflags, int Nf, long long xyz_stride,
                                                          long long uv_stride) {
    // XCD-aware decode: the 8 blocks of one dispatch round-robin group cover 8 different pixel tiles, and
    // successive groups walk the hypothesis tiles of those same pixel tiles.
    const int b = blockIdx.x;
    const int q = b >> 3;
    // order: NT > 0 -> hypothesis tiles innermost (same pixel tile back to back);  NT < 0 -> pixel tiles
    // innermost (the resident blocks write a contiguous band of error-image rows), |NT| hypothesis tiles.
    int ht, pt;
    if (NT > 0) { ht = q % NT; pt = (q / NT) * 8 + (b & 7); }
    else { const int PTG = (PT + 7) >> 3; ht = q / PTG; pt = (q % PTG) * 8 + (b & 7); }
    if (pt >= PT) return;
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x;
    {   // frame batch: all hypotheses of this tile score the same frame (Nf is a multiple of HT)
        const int frame = h0 / Nf;
        xyz += (long long)frame * xyz_stride;
        if (UV) uv += (long long)frame * uv_stride;
    }

    __shared__ __attribute__((aligned(16))) float s_pose[HT * POSE_STRIDE];
    __shared__ float s_red[SOFT ? (K2_THREADS / 64) * HT : 1];
    if (!SPOSE) {
        for (int i = tid; i < nh * POSE_STRIDE; i += K2_THREADS) s_pose[i] = staged[(size_t)h0 * POSE_STRIDE + i];
    }

    const int p0 = (pt * K2_THREADS + tid) * PX;
    const bool valid = p0 < P;  // PX == 4 is only launched with P % 4 == 0
    float X[PX], Y[PX], Z[PX], pu[PX], pv[PX];
    if (PX == 4) {
        if (valid) {
            const f4* src = reinterpret_cast<const f4*>(xyz + (size_t)p0 * 3);
            const f4 a = src[0], bb = src[1], c = src[2];
            X[0] = a.x; Y[0] = a.y; Z[0] = a.z;
            X[1] = a.w; Y[1] = bb.x; Z[1] = bb.y;
            X[2] = bb.z; Y[2] = bb.w; Z[2] = c.x;
            X[3] = c.y; Y[3] = c.z; Z[3] = c.w;
            if (UV) {
                const f4* su = reinterpret_cast<const f4*>(uv + (size_t)p0 * 2);
                const f4 u0 = su[0], u1 = su[1];
                pu[0] = u0.x - cx; pv[0] = u0.y - cy; pu[1] = u0.z - cx; pv[1] = u0.w - cy;
                pu[2] = u1.x - cx; pv[2] = u1.y - cy; pu[3] = u1.z - cx; pv[3] = u1.w - cy;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PX; k++) { X[k] = Y[k] = 0.f; Z[k] = 1.f; pu[k] = pv[k] = 0.f; }
        }
    } else {
        if (valid) {
            X[0] = xyz[(size_t)p0 * 3]; Y[0] = xyz[(size_t)p0 * 3 + 1]; Z[0] = xyz[(size_t)p0 * 3 + 2];
            if (UV) { pu[0] = uv[(size_t)p0 * 2] - cx; pv[0] = uv[(size_t)p0 * 2 + 1] - cy; }
        } else { X[0] = Y[0] = 0.f; Z[0] = 1.f; pu[0] = pv[0] = 0.f; }
    }
    if (!UV) {
#pragma unroll
        for (int k = 0; k < PX; k++) {
            const int p = p0 + k;
            const int y = p / W, x = p - y * W;
            pu[k] = (float)x - cx;
            pv[k] = (float)y - cy;
        }
    }
    if (!SPOSE) __syncthreads();

    float* erow = ERR ? err + (size_t)h0 * P + p0 : nullptr;
    const int wave = tid >> 6, lane = tid & 63;

#pragma unroll 4
    for (int h = 0; h < HT; h++) {
        if (h < nh) {
            f4 r0, r1, r2;
            if (SPOSE) {
                const f4* sp = reinterpret_cast<const f4*>(staged + (size_t)(h0 + h) * POSE_STRIDE);
                r0 = sp[0]; r1 = sp[1]; r2 = sp[2];
            } else {
                const f4* sp = reinterpret_cast<const f4*>(s_pose + h * POSE_STRIDE);
                r0 = sp[0]; r1 = sp[1]; r2 = sp[2];
            }
            float e[PX];
            if (PX % 2 == 0) {
#pragma unroll
                for (int k = 0; k < PX; k += 2) {
                    const f2 e2 = residual2(r0, r1, r2, f2{X[k], X[k + 1]}, f2{Y[k], Y[k + 1]}, f2{Z[k], Z[k + 1]}, f2{pu[k], pu[k + 1]},
                                            f2{pv[k], pv[k + 1]}, clampv);
                    e[k] = e2.x; e[k + 1] = e2.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < PX; k++) e[k] = residual(r0, r1, r2, X[k], Y[k], Z[k], pu[k], pv[k], clampv);
            }
            if (ERR && valid) {
                if (PX == 4) {
                    f4 o = {e[0], e[1], e[2], e[3]};
                    if (kflags & 1) *reinterpret_cast<f4*>(erow + (size_t)h * P) = o; else __builtin_nontemporal_store(o, reinterpret_cast<f4*>(erow + (size_t)h * P));
                } else {
                    __builtin_nontemporal_store(e[0], erow + (size_t)h * P);
                }
            }
            if (SOFT) {
                float s = 0.f;
                if (PX % 2 == 0) {
                    f2 s2 = splat(0.f);
#pragma unroll
                    for (int k = 0; k < PX; k += 2) s2 += soft_inlier2(f2{e[k], e[k + 1]}, kA, kB);
                    s = s2.x + s2.y;
                } else {
#pragma unroll
                    for (int k = 0; k < PX; k++) s += soft_inlier(e[k], kA, kB);
                }
                s = wave_sum(valid ? s : 0.f);
                if (lane == 0) s_red[wave * HT + h] = s;
            }
        }
    }

    if (SOFT) {
        __syncthreads();
        if (tid < nh) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < K2_THREADS / 64; w++) s += s_red[w * HT + tid];
            soft_part[(size_t)pt * N + h0 + tid] = s;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// K2, matrix-core form.  The rigid transform of all pixels under all hypotheses is the product
// [3N x 4] (rows R_i | t_i of every pose) x [4 x P] (X, Y, Z, 1 of every pixel) -- a GEMM with K = 4, which is
// exactly one v_mfma_f32_16x16x4_f32 per 16 x 16 output tile (fp32 in, fp32 accumulate, bit-equal to an fmaf
// chain).  rocprof showed the all-VALU kernel issue-bound on the vector ALUs (SQ_ACTIVE_INST_VALU ~ every
// SIMD cycle) with HBM at ~63 %; the 9 FMAs per (hypothesis, pixel) move to the matrix pipe, which runs
// concurrently with the VALU.  (Round 1's operand layout -- rows (hypothesis, x/y/z/pad), one pair per lane -- is gone: the
// hypothesis-pair layout below needs 12 instead of 16 MFMAs per 1024 pairs and half the VALU work; see k_reproject_hp.)
// --------------------------------------------------------------------------------------------------
DM_INLINE float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in every lane
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));  // row_half_mirror
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

// --------------------------------------------------------------------------------------------------
// K2, matrix-core form with HYPOTHESIS-PAIR packed finishing (round 2).  Same GEMM, other operand roles: per group of 16 hypotheses
// three A operands hold their x-, y- and z-rows (A_x[i][k] = fx R0[k] | fx t0 of hypothesis i, ...), B is unchanged (X, Y, Z, 1 of the
// pixel columns).  D_x / D_y / D_z of one v_mfma_f32_16x16x4_f32 each then give lane (g, c) the SAME component of FOUR hypotheses
// (4g .. 4g+3) for pixel column c in four adjacent registers, so every finishing operation runs packed over a hypothesis pair
// (v_pk_fma_f32 / v_pk_mul_f32: du, dv, du^2 + dv^2, the sigmoid's affine map and 1 + 2^t), the pixel position is a broadcast operand,
// and there is no padding row: 12 MFMAs per 1024 (hypothesis, pixel) pairs instead of 16, ~5 plain VALU operations per pair instead of
// ~11 (the 4 transcendentals per pair -- rcp z, sqrt, exp2, rcp -- are what is left).  With m = 0..3 as before (column c -> pixel
// 4c + m) a lane ends up with 4 consecutive pixels of each of its 4 hypotheses: four dwordx4 stores, each wave store still 4 rows x 256 B.
// Arithmetic per pair is the VALU kernel's (fma(dv, dv, du*du)), the transform is the exact fp32 MFMA.
// --------------------------------------------------------------------------------------------------
// One 64-pixel chunk of 16 hypotheses: 12 MFMAs, then three phases on the VALU.
//   1. packed over HYPOTHESIS pairs (adjacent MFMA output registers): iz = 1 / zc, du = pu + nx iz, dv = pv + ny iz, q = du^2 + dv^2.
//      nx / ny are the NEGATED camera-frame x / y (the A rows carry the sign, so that no negation is left in the loop).
//   2. e[r][m] = min(sqrt(q), clamp): scalar operations, which write each result straight into the register it is stored from
//      (ev[r] = 4 consecutive pixels of hypothesis r, one dwordx4) -- the 4 x 4 transpose between the two packings costs nothing.
//   3. soft-inlier sigmoid packed over PIXEL pairs of one hypothesis, accumulated per hypothesis.
// Z == 0 detection of the fast path: zacc accumulates iz^2 -- rcp(0) = inf sticks (as would a NaN), anything finite stays finite unless
// |z| < 5e-20, which only sends the wave down the (always exact) slow path once more.
// The low parts of the pose records through the matrix core as well (k2_flags bit 27): per 1 024 pairs twelve v_mfma_f32_16x16x16_f16 chained through
// the accumulator of the fp32 ones, D = A_lo 2^16 . B 2^-16 + (A_hi . B).  Only the k = 0..3 slice of the 16-deep fp16 product is used (the operands of the
// lanes 16..63 are zero).  What it removes is the SYSTEMATIC part of the fast form's error -- the fp32 rounding of a hypothesis' record shifts all of its
// projections the same way, which is 96 % of its score error; the correction (<= half an ulp of the record times the coordinate)
// is often below the ulp of the sum it is added to, and survives in expectation, which is all a sum over 300 000 cells needs.
struct LoOps { h4 x, y, z; };
template <bool EXACT_Z, bool SOFT, bool LO = false>
DM_INLINE bool hp_chunk(float ax, float ay, float az, const float (&Bm)[4], const f2 (&ppix)[4], float clampv, float kA, float kB, f4 (&ev)[4],
                        f2 (&sloc)[4], const LoOps* lo = nullptr, const h4* B16 = nullptr) {
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    f2 qq[4][2];
    f2 zacc = splat(0.f);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        // LO: the correction FIRST, as the accumulator the fp32 products are added onto.  The other order loses it: a finished fp32 sum sits on its grid, and
        // adding less than half an ulp returns the same number every time (measured: only 60 % of the error gone); added to the exact first product inside
        // the fused multiply-add it shifts the value BEFORE the rounding, which then keeps it in expectation
        f4 cx4 = z4, cy4 = z4, cz4 = z4;
        if (LO) {
            cx4 = __builtin_amdgcn_mfma_f32_16x16x16f16(lo->x, B16[m], z4, 0, 0, 0);
            cy4 = __builtin_amdgcn_mfma_f32_16x16x16f16(lo->y, B16[m], z4, 0, 0, 0);
            cz4 = __builtin_amdgcn_mfma_f32_16x16x16f16(lo->z, B16[m], z4, 0, 0, 0);
        }
        const f4 nx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, Bm[m], cx4, 0, 0, 0);
        const f4 ny = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, Bm[m], cy4, 0, 0, 0);
        const f4 dz = __builtin_amdgcn_mfma_f32_16x16x4f32(az, Bm[m], cz4, 0, 0, 0);
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            const f2 x = pr ? f2{nx.z, nx.w} : f2{nx.x, nx.y};
            const f2 y = pr ? f2{ny.z, ny.w} : f2{ny.x, ny.y};
            const f2 z = pr ? f2{dz.z, dz.w} : f2{dz.x, dz.y};
            f2 iz = {__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
            if (EXACT_Z) {  // projectPoints: z = Z ? 1/Z : 1
                iz.x = (z.x == 0.0f) ? 1.0f : iz.x;
                iz.y = (z.y == 0.0f) ? 1.0f : iz.y;
            } else {
                zacc = pk_fma(iz, iz, zacc);
            }
            const f2 du = pk_fma(x, iz, splat(ppix[m].x));
            const f2 dv = pk_fma(y, iz, splat(ppix[m].y));
            qq[m][pr] = pk_fma(dv, dv, du * du);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pr = r >> 1;
        ev[r].x = fminf(__builtin_amdgcn_sqrtf((r & 1) ? qq[0][pr].y : qq[0][pr].x), clampv);
        ev[r].y = fminf(__builtin_amdgcn_sqrtf((r & 1) ? qq[1][pr].y : qq[1][pr].x), clampv);
        ev[r].z = fminf(__builtin_amdgcn_sqrtf((r & 1) ? qq[2][pr].y : qq[2][pr].x), clampv);
        ev[r].w = fminf(__builtin_amdgcn_sqrtf((r & 1) ? qq[3][pr].y : qq[3][pr].x), clampv);
        if (SOFT) sloc[r] = soft_inlier2(f2{ev[r].x, ev[r].y}, kA, kB) + soft_inlier2(f2{ev[r].z, ev[r].w}, kA, kB);
    }
    return EXACT_Z ? false : !(zacc.x + zacc.y <= 3.0e38f);
}

// The exact-transform chunk (round 6, k2_flags bit 28): hp_chunk with E = R.X + t from the split records (k_pose_prep_split) -- per row one K = 32 fp16 MFMA
// for the cross terms, one K = 16 fp16 MFMA for the exactly summed high products (its B operand is the first half of the other's), one packed fma per
// register pair for E = D_hi + 2^-14 D_cross: the camera-frame point is rounded to float ONCE.  Then hp_chunk's tail with one Newton step on v_rcp_f32:
// the hardware reciprocal's error is not zero-mean, and a biased 1 / z scales every projection of a hypothesis about the principal point the same way --
// measured (scripts/r06_k2_diag.py, profiles/r06_k2_diag.txt): with an exact E and the plain v_rcp_f32 the softmax weight of two unrelated hypotheses in a tie is
// still off by 5.7e-4 (stated 1e-4), with the Newton step by 5.3e-5; max |err - oracle| over all cells 7.6e-5 px.
struct ExOps { h8 cx, cy, cz; h4 hx, hy, hz; };
// E = D_hi + 2^-14 D_cross on a register pair of the two accumulators.  (Spelled out as inline assembly -- the compiler emits half of these as two
// v_fma_f32 each -- it returned garbage: the hazard recogniser does not see an asm statement's read of a matrix-core result, and gfx950 has no interlock there.)
DM_INLINE f2 ex_combine(f2 c, f2 k, f2 h) { return pk_fma(c, k, h); }
// RSQ (EXF = 2): the tail without the reciprocal of z.  With nu = pu z - xc, nv = pv z - yc (one fma each on the once-rounded point):
//   e = sqrt(nu^2 + nv^2) / |z| = n rsq(n z^2),  n = nu^2 + nv^2
// -- ONE transcendental for reciprocal, Newton step and square root (3 instead of 4 per pair with the sigmoid's two; transcendentals are a third of the
// kernel's issue cycles).  v_rsq_f32's error does not bias a hypothesis against another the way v_rcp_f32's did: its argument n z^2 spreads over many octaves
// within every hypothesis (the reciprocal's argument z sits within two), so what is systematic in it is common to all.  z == 0 (and an exactly zero residual):
// the argument is 0, rsq = inf, which sticks in the accumulated products and sends the chunk down the exact-z path as before.
template <bool EXACT_Z, bool SOFT, bool RSQ = false>
DM_INLINE bool hp_chunk_ex(const ExOps& A, const h8 (&B8)[4], const f2 (&ppix)[4], float clampv, float kA, float kB, f4 (&ev)[4], f2 (&sloc)[4]) {
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f2 k10 = splat(6.103515625e-05f);  // 2^-14: the cross group's scale
    constexpr bool TAIL_RSQ = RSQ && !EXACT_Z;
    f2 qq[4][2];
    f2 zacc = splat(0.f);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const h4 b4 = __builtin_shufflevector(B8[m], B8[m], 0, 1, 2, 3);
        const f4 cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.cx, B8[m], z4, 0, 0, 0);
        const f4 cy = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.cy, B8[m], z4, 0, 0, 0);
        const f4 cz = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.cz, B8[m], z4, 0, 0, 0);
        const f4 hx = __builtin_amdgcn_mfma_f32_16x16x16f16(A.hx, b4, z4, 0, 0, 0);
        const f4 hy = __builtin_amdgcn_mfma_f32_16x16x16f16(A.hy, b4, z4, 0, 0, 0);
        const f4 hz = __builtin_amdgcn_mfma_f32_16x16x16f16(A.hz, b4, z4, 0, 0, 0);
        f2 iz0 = splat(0.f);
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            const f2 x = ex_combine(pr ? f2{cx.z, cx.w} : f2{cx.x, cx.y}, k10, pr ? f2{hx.z, hx.w} : f2{hx.x, hx.y});
            const f2 y = ex_combine(pr ? f2{cy.z, cy.w} : f2{cy.x, cy.y}, k10, pr ? f2{hy.z, hy.w} : f2{hy.x, hy.y});
            const f2 z = ex_combine(pr ? f2{cz.z, cz.w} : f2{cz.x, cz.y}, k10, pr ? f2{hz.z, hz.w} : f2{hz.x, hz.y});
            if (TAIL_RSQ) {
                const f2 nu = pk_fma(splat(ppix[m].x), z, x), nv = pk_fma(splat(ppix[m].y), z, y);  // x, y carry -xc, -yc
                const f2 n = pk_fma(nv, nv, nu * nu);
                const f2 arg = n * (z * z);
                const f2 r = {__builtin_amdgcn_rsqf(arg.x), __builtin_amdgcn_rsqf(arg.y)};
                if (pr == 0) iz0 = r; else zacc = pk_fma(iz0, r, zacc);
                qq[m][pr] = n * r;  // the distance itself
                continue;
            }
            f2 iz = {__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
            iz = pk_fma(pk_fma(-z, iz, splat(1.0f)), iz, iz);  // z == 0: inf -> NaN, which sticks in zacc like the inf of the plain form
            if (EXACT_Z) {  // projectPoints: z = Z ? 1/Z : 1
                iz.x = (z.x == 0.0f) ? 1.0f : iz.x;
                iz.y = (z.y == 0.0f) ? 1.0f : iz.y;
            } else if (pr == 0) {
                iz0 = iz;
            } else {
                zacc = pk_fma(iz0, iz, zacc);  // one fma for both register pairs of an m: a NaN or an inf in either survives the product (inf x 0 = NaN)
            }
            const f2 du = pk_fma(x, iz, splat(ppix[m].x));
            const f2 dv = pk_fma(y, iz, splat(ppix[m].y));
            qq[m][pr] = pk_fma(dv, dv, du * du);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pr = r >> 1;
        const float q0 = (r & 1) ? qq[0][pr].y : qq[0][pr].x, q1 = (r & 1) ? qq[1][pr].y : qq[1][pr].x;
        const float q2 = (r & 1) ? qq[2][pr].y : qq[2][pr].x, q3 = (r & 1) ? qq[3][pr].y : qq[3][pr].x;
        ev[r].x = fminf(TAIL_RSQ ? q0 : __builtin_amdgcn_sqrtf(q0), clampv);
        ev[r].y = fminf(TAIL_RSQ ? q1 : __builtin_amdgcn_sqrtf(q1), clampv);
        ev[r].z = fminf(TAIL_RSQ ? q2 : __builtin_amdgcn_sqrtf(q2), clampv);
        ev[r].w = fminf(TAIL_RSQ ? q3 : __builtin_amdgcn_sqrtf(q3), clampv);
        if (SOFT) sloc[r] = soft_inlier2(f2{ev[r].x, ev[r].y}, kA, kB) + soft_inlier2(f2{ev[r].z, ev[r].w}, kA, kB);
    }
    return EXACT_Z ? false : !(fabsf(zacc.x + zacc.y) <= 3.0e38f);  // products of two reciprocals: -inf counts as well
}

// The B operand of hp_chunk_ex for one (chunk, m): lane (g, c) holds coordinate g of pixel 4c + m (g = 3: the constants the translation pieces multiply).
// X = XT + XA' + XM + XL1 + XL2 (k_pose_prep_split); every remainder below is exact in fp32.  oor: |X| >= 2^16 mm (or NaN).
// Conversions round to nearest (v_cvt_pk_f16_f32): every piece but XL2 is exactly representable, and XL2 must not be TRUNCATED -- the budget for a systematic
// error of the camera-frame point is ~1e-6 mm (k_pose_prep_split).
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
DM_INLINE unsigned pk_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2v)); }
// The lanes of quarter 3 (the translation) run the same code on the constant SPLIT_T_COORD = 2^15 + 1, whose pieces are the powers of two
// (2^10, 2^3, 2^11, 0, 0, 0, 1, 0) that the translation pieces of k_pose_prep_split are scaled against -- no select between coordinates and constants.
constexpr float SPLIT_T_COORD = 32769.0f;
DM_INLINE h8 split_B(float X, bool& oor) {
    const float kt = __builtin_rintf(X * 0.0009765625f);      // XT / 1024
    const float r1 = fmaf(-1024.f, kt, X);                     // X - XT, |.| <= 512
    const float ka = __builtin_rintf(r1 * 0.03125f);           // XA' / 32
    const float r2 = fmaf(-32.f, ka, r1);                      // X - XA, |.| <= 16
    const float xm = __builtin_rintf(r2);                      // XM
    const float xl = r2 - xm;                                  // |.| <= 0.5
    const float kl = __builtin_rintf(xl * 4096.f);             // XL1 2^12
    const float xl2 = fmaf(-0.000244140625f, kl, xl);          // XL2
    const float ja = fmaf(32.f, kt, ka);                       // XA / 32
    oor = oor || !(fabsf(ja) <= 2047.f);
    const u4v v = {pk_h2(ja, xm * 8.f), pk_h2(kt * 64.f, kl * 0.25f), pk_h2(xl2 * 1024.f, ka * 32.f), pk_h2(xm, kl * 0.000244140625f)};
    return __builtin_bit_cast(h8, v);
}

template <int HT, bool ERR, bool SOFT, bool UV, int KM_CH>
__global__ __launch_bounds__(K2_THREADS) void k_reproject_hp(const float* __restrict__ staged, const float* __restrict__ xyz,
                                                             const float* __restrict__ uv, float* __restrict__ err,
                                                             float* __restrict__ soft_part, int N, int P, int W, int PT, int NT, float cx,
                                                             float cy, float clampv, float kA, float kB, int kflags, int Nf,
                                                             long long xyz_stride, long long uv_stride) {
    static_assert(HT % 16 == 0, "hypothesis tile must be a multiple of 16");
    const int b = blockIdx.x;
    const int q = b >> 3;
    int ht, pt;  // block order as in k_reproject (NT > 0: hypothesis tiles innermost; NT < 0: pixel tiles innermost)
    if (NT > 0) { ht = q % NT; pt = (q / NT) * 8 + (b & 7); }
    else { const int PTG = (PT + 7) >> 3; ht = q / PTG; pt = (q % PTG) * 8 + (b & 7); }
    if (pt >= PT) return;
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    {   // frame batch: all hypotheses of this tile score the same frame (Nf is a multiple of HT)
        const int frame = h0 / Nf;
        xyz += (long long)frame * xyz_stride;
        if (UV) uv += (long long)frame * uv_stride;
    }

    // A operands in MFMA lane order: s_A[(gi*3 + comp)*64 + l] = record[h0 + 16 gi + l%16][comp][k = l/16]; hypotheses beyond the
    // ragged end repeat the last valid one (their results are never stored, and a zero record would send every wave down the Z == 0 path)
    __shared__ float s_A[(HT / 16) * 3 * 64];
    __shared__ float s_soft[SOFT ? (K2_THREADS / 64) * HT : 1];
    for (int i = tid; i < (HT / 16) * 3 * 64; i += K2_THREADS) {
        const int l = i & 63, cg = i >> 6;
        const int comp = cg % 3, gi = cg / 3;
        const int hyp = min(16 * gi + (l & 15), nh - 1), k = l >> 4;
        const float v = staged[(size_t)(h0 + hyp) * POSE_STRIDE + comp * 4 + k];
        s_A[i] = comp < 2 ? -v : v;  // x and y rows negated: the MFMA yields (-xc, -yc, zc), du = pu + (-xc) / zc needs no sign flip
    }

    float Bm[KM_CH][4];
    f2 ppix[KM_CH][4];
    int p0[KM_CH];
    bool valid[KM_CH];
#pragma unroll
    for (int ch = 0; ch < KM_CH; ch++) {
        const int chunk0 = (pt * (K2_THREADS / 64) * KM_CH + wave * KM_CH + ch) * 64;
        p0[ch] = chunk0 + 4 * c;       // this lane's 4 consecutive pixels
        valid[ch] = p0[ch] < P;        // P % 4 == 0
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int pc = min(p0[ch] + m, P - 1);
            Bm[ch][m] = (g < 3) ? xyz[(size_t)pc * 3 + g] : 1.0f;
        }
        if (UV) {
            if (valid[ch]) {
                const f4* su = reinterpret_cast<const f4*>(uv + (size_t)p0[ch] * 2);
                const f4 u0 = su[0], u1 = su[1];
                ppix[ch][0] = f2{u0.x - cx, u0.y - cy}; ppix[ch][1] = f2{u0.z - cx, u0.w - cy};
                ppix[ch][2] = f2{u1.x - cx, u1.y - cy}; ppix[ch][3] = f2{u1.z - cx, u1.w - cy};
            } else {
                ppix[ch][0] = ppix[ch][1] = ppix[ch][2] = ppix[ch][3] = splat(0.f);
            }
        } else {
            int y = p0[ch] / W, x = p0[ch] - y * W;  // implicit grid: one integer division per chunk, then walk (with row wrap)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                ppix[ch][m] = f2{(float)x - cx, (float)y - cy};
                if (++x == W) { x = 0; y++; }
            }
        }
    }
    __syncthreads();

    for (int gi = 0; gi < HT / 16; gi++) {
        if (16 * gi >= nh) break;
        const float ax = s_A[(gi * 3 + 0) * 64 + lane], ay = s_A[(gi * 3 + 1) * 64 + lane], az = s_A[(gi * 3 + 2) * 64 + lane];
        const int hyp0 = 16 * gi + 4 * g;  // this lane's 4 hypotheses: hyp0 .. hyp0 + 3
        f2 ssum[4] = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
#pragma unroll
        for (int ch = 0; ch < KM_CH; ch++) {
            f4 ev[4];  // ev[r] = 4 consecutive pixels of hypothesis hyp0 + r
            f2 sloc[4];
            if (kflags & 2) {  // measurement knob (k2_flags bit 1): the store schedule alone, no arithmetic -- the ceiling of this write pattern
                if (ERR && valid[ch]) {
                    f4* dst = reinterpret_cast<f4*>(err + (size_t)(h0 + hyp0) * P + p0[ch]);
                    const size_t rs = (size_t)P / 4;
                    const f4 o = {ax, ay, az, (float)gi};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (hyp0 + r >= nh) continue;
                        f4* q = dst + r * rs;
                        const int mode = (kflags >> 2) & 7;  // cache policy of the store (measurement): 0 nt, 1 plain, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 sc0
                        if (mode == 0) __builtin_nontemporal_store(o, q);
                        else if (mode == 1) *q = o;
                        else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(o) : "memory");
                        else if (mode == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(o) : "memory");
                        else if (mode == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(q), "v"(o) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(q), "v"(o) : "memory");
                    }
                }
                continue;
            }
            // projectPoints' "z = Z ? 1/Z : 1": one test for the 16 pairs, wave-uniform slow path (practically never) that redoes the chunk
            if (__builtin_expect(__any(hp_chunk<false, SOFT>(ax, ay, az, Bm[ch], ppix[ch], clampv, kA, kB, ev, sloc)), 0))
                (void)hp_chunk<true, SOFT>(ax, ay, az, Bm[ch], ppix[ch], clampv, kA, kB, ev, sloc);
            if (SOFT) {
                const f2 vf = splat(valid[ch] ? 1.0f : 0.0f);
#pragma unroll
                for (int r = 0; r < 4; r++) ssum[r] = pk_fma(sloc[r], vf, ssum[r]);
            }
            if (ERR && valid[ch]) {
                f4* dst = reinterpret_cast<f4*>(err + (size_t)(h0 + hyp0) * P + p0[ch]);
                const size_t rs = (size_t)P / 4;  // row stride in f4 (P % 4 == 0)
                if (nh == HT) {  // full tile (uniform): no per-row guards
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (kflags & 1) dst[r * rs] = ev[r]; else __builtin_nontemporal_store(ev[r], dst + r * rs);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (hyp0 + r < nh) { if (kflags & 1) dst[r * rs] = ev[r]; else __builtin_nontemporal_store(ev[r], dst + r * rs); }
                    }
                }
            }
        }
        if (SOFT) {
            // the 16 lanes of a DPP row hold the same 4 hypotheses
            const float s0 = row16_sum(ssum[0].x + ssum[0].y), s1 = row16_sum(ssum[1].x + ssum[1].y);
            const float s2 = row16_sum(ssum[2].x + ssum[2].y), s3 = row16_sum(ssum[3].x + ssum[3].y);
            if (c == 0) {  // written exactly once per (wave, hypothesis)
                float* dst = s_soft + wave * HT + hyp0;
                if (hyp0 + 0 < nh) dst[0] = s0;
                if (hyp0 + 1 < nh) dst[1] = s1;
                if (hyp0 + 2 < nh) dst[2] = s2;
                if (hyp0 + 3 < nh) dst[3] = s3;
            }
        }
    }

    if (SOFT) {
        __syncthreads();
        if (tid < nh) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < K2_THREADS / 64; w++) s += s_soft[w * HT + tid];
            soft_part[(size_t)pt * N + h0 + tid] = s;
        }
    }
}

template <int HT, int KM_CH>
static hipError_t launch_reproject_hp(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float kA,
                                      float kB, float* soft_part, int* tiles_used, int Nf, bool pixel_minor, int kflags, hipEvent_t evA, hipEvent_t evB) {
    const int tile = K2_THREADS * KM_CH;  // 64 pixels per wave-chunk
    const int PT = (F.P + tile - 1) / tile;
    const int NTa = (N + HT - 1) / HT;
    const int grid = ((PT + 7) / 8) * 8 * NTa;
    const int NT = pixel_minor ? -NTa : NTa;
    if (tiles_used) *tiles_used = PT;
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
    // experiment knob (k2_flags bits 8..15): that many 8-KiB units of unused dynamic LDS per workgroup cap the workgroups per CU
    const size_t lds_pad = (size_t)((kflags >> 8) & 0xff) * 8192;
#define DSAC_K2H(E, S, U)                                                                                                             \
    hipExtLaunchKernelGGL((k_reproject_hp<HT, E, S, U, KM_CH>), dim3(grid), dim3(K2_THREADS), lds_pad, st, evA, evB, 0, staged, F.xyz, F.uv, err,   \
                          soft_part, N, F.P, F.W, PT, NT, F.cx, F.cy, clampv, kA, kB, kflags, Nf, F.xyz_stride, F.uv_stride)
    if (ERR && SOFT) { if (UV) DSAC_K2H(true, true, true); else DSAC_K2H(true, true, false); }
    else if (ERR) { if (UV) DSAC_K2H(true, false, true); else DSAC_K2H(true, false, false); }
    else if (SOFT) { if (UV) DSAC_K2H(false, true, true); else DSAC_K2H(false, true, false); }
#undef DSAC_K2H
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K2, streaming small-tile form (round 3).  Same arithmetic as k_reproject_hp (hp_chunk), other schedule.
// What decides the store rate of the N x P volume is how many error-image ROWS are open at once across the chip
// (scripts/micro/store_sched.hip, profiles/r03_store_sched.txt: one row 6.9 TB/s, ~30 rows 6.6, ~110 rows 5.8, ~450 rows 5.5):
//   open rows ~ hypotheses per workgroup x (resident waves x pixels a wave holds while it lives) / P.
// So the tile is small (NG x 16 hypotheses, WAVES x CHW x 64 pixels) and its workgroups short-lived, and the per-workgroup set-up
// that made round 2's small tiles lose is cut down: the MFMA A operands come straight from the staged records (one dword per lane and
// component, L2-resident -- no LDS image, no barrier before the first MFMA), the pixel positions of the implicit grid from a
// wave-uniform row / column (W % 64 == 0), and the only barrier is the one before the 16 NG partial soft sums leave the workgroup.
// --------------------------------------------------------------------------------------------------
template <int NG, int CHW, int WAVES, bool PW, bool ERR, bool SOFT, bool UV, bool G64, int MINW, bool LO = false, int EXF = 0>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_reproject_st(const float* __restrict__ staged, const float* __restrict__ xyz,
                                                             const float* __restrict__ uv, float* __restrict__ err,
                                                             float* __restrict__ soft_part, int N, int P, int W, int PT, float cx, float cy,
                                                             float clampv, float kA, float kB, int kflags, int Nf, long long xyz_stride,
                                                             long long uv_stride, const float* __restrict__ staged_lo = nullptr,
                                                             const void* __restrict__ split = nullptr) {
    constexpr int HT = 16 * NG;
    constexpr bool EX = EXF != 0, RSQ = EXF == 2;  // EXF: 0 = fp32 transform, 1 = exact transform (split fp16 records), 2 = the same with the one-transcendental tail (hp_chunk_ex)
    const int b = blockIdx.x;
    int ht, pt;
    if (kflags & 32) { pt = b % PT; ht = b / PT; }  // plain pixel-minor order
    else if (WAVES == 1) {
        // one-wave workgroups, XCD-aware: the four workgroups b, b + 8, b + 16, b + 24 (same XCD, dispatched back to back) take four ADJACENT
        // 64-pixel chunks, so that an XCD still writes 1 KiB runs of every row
        const int q = b >> 5, sub = (b >> 3) & 3, PTG = (PT + 31) >> 5;
        ht = q / PTG;
        pt = ((q % PTG) * 8 + (b & 7)) * 4 + sub;
        if (pt >= PT) return;
    } else {  // XCD-aware pixel-minor order (block b -> XCD b % 8 keeps its eighth of the pixel tiles)
        const int q = b >> 3, PTG = (PT + 7) >> 3;
        ht = q / PTG;
        pt = (q % PTG) * 8 + (b & 7);
        if (pt >= PT) return;
    }
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    {
        const int frame = h0 / Nf;
        xyz += (long long)frame * xyz_stride;
        if (UV) uv += (long long)frame * uv_stride;
    }
    __shared__ float s_soft[(SOFT && !PW) ? WAVES * HT : 1];
    // PW: every wave writes its own partial soft sums (row pt * WAVES + wave of soft_part), no LDS, no barrier

    // B operands (coordinate g of pixel 4c + m of every chunk) and the MFMA A operands (x-, y-, z-row entry k = g of hypothesis c)
    // G64 (implicit grid, W % 64 == 0: a 64-pixel chunk never straddles a row): the pixel positions are rebuilt from one (column, row) pair
    // per chunk in front of every use instead of living in 8 registers per chunk
    float Bm[CHW][4];
    f2 ppix[G64 ? 1 : CHW][4];
    // G64: row and first column of a chunk are wave-uniform and live in SCALAR registers (late round 3: as a lane-dependent pair per chunk they
    // were 8 vector registers, exactly what the 128-register build of the <64 hypotheses, 256 pixels> form spilled)
    int sx0[G64 ? CHW : 1];
    float sby[G64 ? CHW : 1];
    int p0[CHW];
    bool valid[CHW];
    const int chunk0 = (pt * WAVES + wave) * CHW;
#pragma unroll
    for (int ch = 0; ch < CHW; ch++) {
        p0[ch] = (chunk0 + ch) * 64 + 4 * c;
        valid[ch] = p0[ch] < P;  // P % 4 == 0
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int pc = min(p0[ch] + m, P - 1);
            Bm[ch][m] = (g < 3) ? xyz[(size_t)pc * 3 + g] : 1.0f;
        }
    }
    // EX: the split fp16 B operands of every (chunk, m); a chunk with a coordinate beyond 2^16 mm takes the fp32 path (wave-uniform flag per chunk)
    h8 B8[EX ? CHW : 1][4];
    bool oor[EX ? CHW : 1];
    if (EX) {
#pragma unroll
        for (int ch = 0; ch < CHW; ch++) {
            bool o = false;
#pragma unroll
            for (int m = 0; m < 4; m++) B8[ch][m] = split_B((g < 3) ? Bm[ch][m] : SPLIT_T_COORD, o);
            oor[ch] = __any(o);
        }
    }
    // LO: the fp16 B operands of every (chunk, m) -- (X, Y, Z, 1) 2^-16 of pixel column c in the lanes of row 0, zero elsewhere.  Row 0 holds X itself;
    // Y and Z come over from rows 1 and 2 with one lane swap each
    h4 B16[LO ? CHW : 1][4];
    if (LO) {
#pragma unroll
        for (int ch = 0; ch < CHW; ch++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const unsigned v = __builtin_bit_cast(unsigned, Bm[ch][m]);
                const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // [1]: row 0 <- row 1
                const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // [1]: rows 0, 1 <- rows 2, 3
                const unsigned u16 = r16[1], u32 = r32[1];
                const float s = 1.52587890625e-05f;  // 2^-16
                const float X = Bm[ch][m] * s, Y = __builtin_bit_cast(float, u16) * s, Z = __builtin_bit_cast(float, u32) * s;
                const bool r0 = g == 0;
                // the constant 1 of the translation column travels as 2^-8 (its record entry as lo 2^8): 2^-16 is a SUBNORMAL half, which the matrix core
                // flushes -- measured: with it the translation's correction, the largest of the four, was simply missing
                B16[ch][m] = h4{(_Float16)(r0 ? X : 0.f), (_Float16)(r0 ? Y : 0.f), (_Float16)(r0 ? Z : 0.f), (_Float16)(r0 ? 0.00390625f : 0.f)};
            }
    }
    // A operands: with up to two groups all of them are loaded up front; with four they are fetched one group ahead (two register sets instead of
    // four: the 128-register build of the <64 hypotheses, 256 pixels> form needs the six registers)
    constexpr bool A_AHEAD = NG > 2;
    float ax[EX ? 1 : NG], ay[EX ? 1 : NG], az[EX ? 1 : NG];
    LoOps alo[LO ? NG : 1];
    ExOps aex[EX ? NG : 1];
    auto load_A = [&](int gi) {
        const int hyp = min(16 * gi + c, nh - 1);  // beyond the ragged end: repeat the last valid hypothesis (never stored)
        if (EX) {
            // the split records in the matrix core's lane layout: quarter g of rows x, y, z of hypothesis c -- 16 bytes (K = 32 operand) + 8 bytes (K = 16) each
            const h8* rc = reinterpret_cast<const h8*>(split) + (size_t)(h0 + hyp) * 12 + g;
            const h4* rh = reinterpret_cast<const h4*>(reinterpret_cast<const char*>(split) + (size_t)N * SPLIT_CROSS_BYTES) + (size_t)(h0 + hyp) * 12 + g;
            aex[gi].cx = rc[0]; aex[gi].cy = rc[4]; aex[gi].cz = rc[8];
            aex[gi].hx = rh[0]; aex[gi].hy = rh[4]; aex[gi].hz = rh[8];
            return;
        }
        const float* rec = staged + (size_t)(h0 + hyp) * POSE_STRIDE + g;
        ax[gi] = rec[0]; ay[gi] = rec[4]; az[gi] = rec[8];
        if (LO) {
            // the records' low parts x 2^16 (x and y rows negated like the high parts), k = 0..3 of hypothesis c in the lanes of row 0
            const f4* rl = reinterpret_cast<const f4*>(staged_lo + (size_t)(h0 + hyp) * POSE_STRIDE);
            const float S = (g == 0) ? 65536.f : 0.f, T = (g == 0) ? 256.f : 0.f;  // rotation entries x 2^16 against coordinates x 2^-16, translation x 2^8 against 2^-8
            const f4 lx = rl[0], ly = rl[1], lz = rl[2];
            alo[gi].x = h4{(_Float16)(-lx.x * S), (_Float16)(-lx.y * S), (_Float16)(-lx.z * S), (_Float16)(-lx.w * T)};
            alo[gi].y = h4{(_Float16)(-ly.x * S), (_Float16)(-ly.y * S), (_Float16)(-ly.z * S), (_Float16)(-ly.w * T)};
            alo[gi].z = h4{(_Float16)(lz.x * S), (_Float16)(lz.y * S), (_Float16)(lz.z * S), (_Float16)(lz.w * T)};
        }
    };
#pragma unroll
    for (int gi = 0; gi < (A_AHEAD ? 1 : NG); gi++) load_A(gi);
#pragma unroll
    for (int ch = 0; ch < CHW; ch++) {
        if (UV) {
            constexpr int pc = G64 ? 0 : 1;
            if (valid[ch]) {
                const f4* su = reinterpret_cast<const f4*>(uv + (size_t)p0[ch] * 2);
                const f4 u0 = su[0], u1 = su[1];
                ppix[ch * pc][0] = f2{u0.x - cx, u0.y - cy}; ppix[ch * pc][1] = f2{u0.z - cx, u0.w - cy};
                ppix[ch * pc][2] = f2{u1.x - cx, u1.y - cy}; ppix[ch * pc][3] = f2{u1.z - cx, u1.w - cy};
            } else {
                ppix[ch * pc][0] = ppix[ch * pc][1] = ppix[ch * pc][2] = ppix[ch * pc][3] = splat(0.f);
            }
        } else if (G64) {
            // row and first column are wave-uniform (scalar division)
            const int cpr = W >> 6;
            const int y = (chunk0 + ch) / cpr, xc0 = ((chunk0 + ch) - y * cpr) * 64;
            sx0[ch] = __builtin_amdgcn_readfirstlane(xc0);
            sby[ch] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)y - cy)));
        } else {
            constexpr int pc = G64 ? 0 : 1;
            int y = p0[ch] / W, x = p0[ch] - y * W;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                ppix[ch * pc][m] = f2{(float)x - cx, (float)y - cy};
                if (++x == W) { x = 0; y++; }
            }
        }
    }

#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
        if (16 * gi >= nh) break;
        if (A_AHEAD && gi + 1 < NG) {
            asm volatile("" ::: "memory");  // keeps the compiler from hoisting the fetch to the top of the kernel (that is the up-front form)
            load_A(gi + 1);
        }
        const float nax = EX ? 0.f : -ax[EX ? 0 : gi], nay = EX ? 0.f : -ay[EX ? 0 : gi];  // x and y rows negated: the MFMA yields (-xc, -yc, zc)
        const int hyp0 = 16 * gi + 4 * g;
        f2 ssum[4] = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
#pragma unroll
        for (int ch = 0; ch < CHW; ch++) {
            f4 ev[4];
            f2 sloc[4];
            if (G64) {
                f2 pb = f2{(float)(sx0[ch] + 4 * c) - cx, sby[ch]};  // the same numbers as a stored pair: (float)(column) - cx, (float)(row) - cy
                asm volatile("" : "+v"(pb));  // rebuilt per use: keeps the 8 position registers of a chunk from staying live across the whole kernel
#pragma unroll
                for (int m = 0; m < 4; m++) ppix[0][m] = f2{pb.x + (float)m, pb.y};
            }
            const f2 (&pp)[4] = ppix[G64 ? 0 : ch];
            if (EX) {
                if (__builtin_expect(oor[ch], 0)) {
                    // a coordinate beyond the split's range: this chunk through the fp32 transform (operands fetched here, the rare path)
                    const int hyp = min(16 * gi + c, nh - 1);
                    const float* rec = staged + (size_t)(h0 + hyp) * POSE_STRIDE + g;
                    float Bf[4];
#pragma unroll
                    for (int m = 0; m < 4; m++) Bf[m] = (g < 3) ? xyz[(size_t)min(p0[ch] + m, P - 1) * 3 + g] : 1.0f;
                    (void)hp_chunk<true, SOFT, false>(-rec[0], -rec[4], rec[8], Bf, pp, clampv, kA, kB, ev, sloc);
                } else if (__builtin_expect(__any(hp_chunk_ex<false, SOFT, RSQ>(aex[gi], B8[ch], pp, clampv, kA, kB, ev, sloc)), 0)) {
                    (void)hp_chunk_ex<true, SOFT>(aex[gi], B8[ch], pp, clampv, kA, kB, ev, sloc);
                }
            } else if (kflags & 2) {  // store schedule alone (measurement)
                ev[0] = ev[1] = ev[2] = ev[3] = f4{nax, nay, az[EX ? 0 : gi], (float)ch};
                sloc[0] = sloc[1] = sloc[2] = sloc[3] = splat(0.f);
            } else if (__builtin_expect(__any(hp_chunk<false, SOFT, LO>(nax, nay, az[EX ? 0 : gi], Bm[ch], pp, clampv, kA, kB, ev, sloc, &alo[LO ? gi : 0], B16[LO ? ch : 0])), 0)) {
                (void)hp_chunk<true, SOFT, LO>(nax, nay, az[EX ? 0 : gi], Bm[ch], pp, clampv, kA, kB, ev, sloc, &alo[LO ? gi : 0], B16[LO ? ch : 0]);
            }
            if (SOFT) {
                const f2 vf = splat(valid[ch] ? 1.0f : 0.0f);
#pragma unroll
                for (int r = 0; r < 4; r++) ssum[r] = pk_fma(sloc[r], vf, ssum[r]);
            }
            if (ERR && valid[ch]) {
                // buffer store: the address is a wave-uniform row base in a scalar resource (the tile's first hypothesis of this group) + a scalar row offset +
                // a 32-bit lane offset (the lane's hypothesis quarter and pixel).  A global store's 64-bit vector address cost one vector instruction per store
                // (v_lshl_add_u64), 16 per 1 024 pairs of a kernel that is bound by vector issue; this form steps the rows on the scalar unit
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(err + (size_t)(h0 + 16 * gi) * P, 0, 0xffffffffu, 0x00020000);
                const unsigned loff = ((unsigned)(4 * g) * (unsigned)P + (unsigned)p0[ch]) * 4u;
                if (nh == HT) {
#pragma unroll
                    for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, ev[r]), rsrc, loff, (unsigned)r * (unsigned)P * 4u, 2);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (hyp0 + r < nh) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, ev[r]), rsrc, loff, (unsigned)r * (unsigned)P * 4u, 2);
                }
            }
            if (ERR && !SOFT) {
                // error images only: the sigmoid arithmetic of the fused form is what spaces a wave's stores; without it the
                // wave idles for as long instead (k2_flags bits 16-19: units of 64 clocks per chunk, bits 20-23: units of 16) -- no VALU issue slots taken
                const int ps = (kflags >> 16) & 15, pn = (kflags >> 20) & 15;
                for (int i = 0; i < ps; i++) __builtin_amdgcn_s_sleep(1);
                for (int i = 0; i < pn; i++) asm volatile("s_nop 15");
            }
        }
        if (SOFT) {
            const float s0 = row16_sum(ssum[0].x + ssum[0].y), s1 = row16_sum(ssum[1].x + ssum[1].y);
            const float s2 = row16_sum(ssum[2].x + ssum[2].y), s3 = row16_sum(ssum[3].x + ssum[3].y);
            if (c == 0) {
                float* dst = !PW ? s_soft + wave * HT + hyp0 : soft_part + (size_t)(pt * WAVES + wave) * N + h0 + hyp0;
                if (hyp0 + 0 < nh) dst[0] = s0;
                if (hyp0 + 1 < nh) dst[1] = s1;
                if (hyp0 + 2 < nh) dst[2] = s2;
                if (hyp0 + 3 < nh) dst[3] = s3;
            }
        }
    }
    if (SOFT && !PW) {
        __syncthreads();
        if (tid < nh) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; w++) s += s_soft[w * HT + tid];
            soft_part[(size_t)pt * N + h0 + tid] = s;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// K2, persistent pipelined form (round 3).  Small tiles (NG x 16 hypotheses x one 64-pixel chunk per wave) stream their stores best, but as
// short-lived workgroups each of them starts with global loads that queue BEHIND the stores of the other waves of its CU in the CU's
// in-order memory pipe, and a store-bound kernel keeps that pipe full (k2_lab: small tiles 774 us store-only, 846 arithmetic-only, 981
// together).  Here the waves are persistent and software-pipelined: the operands of tile i + 2 (4 coordinate dwords + 3 NG pose dwords per
// lane) are requested BEFORE the stores of tile i are issued, so -- the memory counter of a wave retires in order -- waiting for the operands
// of tile i + 1 never waits for a store, and no wave ever waits for its stores at all until the kernel ends.  The tile sequence of the grid
// follows the order the dispatcher would give short-lived workgroups: pixel units (4 adjacent chunks = the 4 waves of a workgroup) innermost,
// XCD x owning the units u % 8 == x.
// --------------------------------------------------------------------------------------------------
template <int NG>
struct K2Ops {
    float B[4];
    float A[NG][3];
};

template <int NG, int WAVES, bool ERR, bool SOFT, bool UV>
__global__ __launch_bounds__(WAVES * 64) void k_reproject_ps(const float* __restrict__ staged, const float* __restrict__ xyz,
                                                             const float* __restrict__ uv, float* __restrict__ err,
                                                             float* __restrict__ soft_part, int N, int P, int W, float cx, float cy,
                                                             float clampv, float kA, float kB, int kflags, int Nf, long long xyz_stride,
                                                             long long uv_stride) {
    constexpr int HT = 16 * NG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, lwg = blockIdx.x >> 3, GX = gridDim.x >> 3;  // gridDim.x is a multiple of 8
    const int CR = (P + 63) >> 6;                   // 64-pixel chunks per row
    const int UR = (CR + WAVES - 1) / WAVES;        // pixel units (WAVES adjacent chunks) per row
    const int URX = (UR + 7) >> 3;                  // ... per XCD
    const int NRG = (N + HT - 1) / HT;              // hypothesis groups
    const int TT = NRG * URX;                       // tiles of this XCD's workgroup sequence

    // tile tw of this workgroup -> (first hypothesis, chunk of this wave); chunk < 0: nothing to do
    auto decode = [&](int tw, int& h0, int& chunk) {
        const int rg = tw / URX, ux = tw - rg * URX;
        const int unit = ux * 8 + xcd;
        h0 = rg * HT;
        chunk = (tw < TT && unit < UR) ? unit * WAVES + wave : -1;
        if (chunk >= CR) chunk = -1;
    };
    // the loads of a tile are issued unconditionally (a tile beyond the end re-reads the last one's operands): the number of memory
    // operations between a tile's loads and their first use is then the same on every path, and the wait in front of that use is never vmcnt(0)
    auto issue = [&](int tw, K2Ops<NG>& o) {
        int h0, chunk;
        decode(min(tw, TT - 1), h0, chunk);
        chunk = max(chunk, 0);
        const int nh = min(HT, N - h0);
        const float* fx = xyz + (long long)(h0 / Nf) * xyz_stride;
        const int p0 = chunk * 64 + 4 * c;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int pc = min(p0 + m, P - 1);
            o.B[m] = fx[(size_t)pc * 3 + min(g, 2)];
        }
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            const int hyp = min(16 * gi + c, nh - 1);
            const float* rec = staged + (size_t)(h0 + hyp) * POSE_STRIDE + g;
            o.A[gi][0] = rec[0]; o.A[gi][1] = rec[4]; o.A[gi][2] = rec[8];
        }
    };
    auto compute = [&](int tw, const K2Ops<NG>& o) {
        int h0, chunk;
        decode(tw, h0, chunk);
        if (chunk < 0) return;
        const int nh = min(HT, N - h0);
        const int p0 = chunk * 64 + 4 * c;
        const bool valid = p0 < P;
        const float Bv[4] = {g < 3 ? o.B[0] : 1.0f, g < 3 ? o.B[1] : 1.0f, g < 3 ? o.B[2] : 1.0f, g < 3 ? o.B[3] : 1.0f};  // the k = 3 row of B is 1
        f2 ppix[4];
        if (UV) {
            const float* fu = uv + (long long)(h0 / Nf) * uv_stride;
            if (valid) {
                const f4* su = reinterpret_cast<const f4*>(fu + (size_t)p0 * 2);
                const f4 u0 = su[0], u1 = su[1];
                ppix[0] = f2{u0.x - cx, u0.y - cy}; ppix[1] = f2{u0.z - cx, u0.w - cy};
                ppix[2] = f2{u1.x - cx, u1.y - cy}; ppix[3] = f2{u1.z - cx, u1.w - cy};
            } else {
                ppix[0] = ppix[1] = ppix[2] = ppix[3] = splat(0.f);
            }
        } else if ((W & 63) == 0) {
            const int cpr = W >> 6;
            const int y = chunk / cpr, xc0 = (chunk - y * cpr) * 64;
            const float fy_ = (float)y - cy, fx_ = (float)(xc0 + 4 * c) - cx;
#pragma unroll
            for (int m = 0; m < 4; m++) ppix[m] = f2{fx_ + (float)m, fy_};
        } else {
            int y = p0 / W, x = p0 - y * W;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                ppix[m] = f2{(float)x - cx, (float)y - cy};
                if (++x == W) { x = 0; y++; }
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            if (16 * gi >= nh) break;
            const float nax = -o.A[gi][0], nay = -o.A[gi][1], az = o.A[gi][2];
            const int hyp0 = 16 * gi + 4 * g;
            f4 ev[4];
            f2 sloc[4];
            if (kflags & 2) {
                ev[0] = ev[1] = ev[2] = ev[3] = f4{nax, nay, az, Bv[0]};
                sloc[0] = sloc[1] = sloc[2] = sloc[3] = splat(0.f);
            } else if (__builtin_expect(__any(hp_chunk<false, SOFT>(nax, nay, az, Bv, ppix, clampv, kA, kB, ev, sloc)), 0)) {
                (void)hp_chunk<true, SOFT>(nax, nay, az, Bv, ppix, clampv, kA, kB, ev, sloc);
            }
            if (ERR && valid) {
                f4* dst = reinterpret_cast<f4*>(err + (size_t)(h0 + hyp0) * P + p0);
                const size_t rs = (size_t)P / 4;
                if (nh == HT) {
#pragma unroll
                    for (int r = 0; r < 4; r++) __builtin_nontemporal_store(ev[r], dst + r * rs);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (hyp0 + r < nh) __builtin_nontemporal_store(ev[r], dst + r * rs);
                }
            }
            if (SOFT) {
                const float vf = valid ? 1.0f : 0.0f;
                const float s0 = row16_sum((sloc[0].x + sloc[0].y) * vf), s1 = row16_sum((sloc[1].x + sloc[1].y) * vf);
                const float s2 = row16_sum((sloc[2].x + sloc[2].y) * vf), s3 = row16_sum((sloc[3].x + sloc[3].y) * vf);
                if (c == 0) {  // one partial row per 64-pixel chunk
                    float* dst = soft_part + (size_t)chunk * N + h0 + hyp0;
                    if (nh == HT) {
                        __builtin_nontemporal_store(f4{s0, s1, s2, s3}, reinterpret_cast<f4*>(dst));
                    } else {
                        if (hyp0 + 0 < nh) dst[0] = s0;
                        if (hyp0 + 1 < nh) dst[1] = s1;
                        if (hyp0 + 2 < nh) dst[2] = s2;
                        if (hyp0 + 3 < nh) dst[3] = s3;
                    }
                }
            }
        }
    };

    // three operand sets in rotation (unrolled by hand: a register copy of a set whose loads are still in flight would have to wait for them)
    K2Ops<NG> o0, o1, o2;
    issue(lwg, o0);
    issue(lwg + GX, o1);
    for (int tw = lwg; tw < TT; tw += 3 * GX) {
        issue(tw + 2 * GX, o2);
        compute(tw, o0);
        issue(tw + 3 * GX, o0);
        compute(tw + GX, o1);
        issue(tw + 4 * GX, o1);
        compute(tw + 2 * GX, o2);
    }
}

template <int NG, int WAVES>
static hipError_t launch_reproject_ps(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float kA, float kB,
                                      float* soft_part, int* tiles_used, int Nf, int kflags, hipEvent_t evA, hipEvent_t evB) {
    if (N % 4 != 0 && soft_part) return hipErrorInvalidValue;  // the partial sums leave as 16-byte stores
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int per_cu = (kflags >> 8) & 0xff;  // k2_flags bits 8..15: persistent workgroups per CU (0 = default)
    if (per_cu <= 0) per_cu = 3;
    const int grid = ((cus * per_cu + 7) / 8) * 8;
    if (tiles_used) *tiles_used = (F.P + 63) / 64;
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
#define DSAC_K2P(E, S, U)                                                                                                               \
    hipExtLaunchKernelGGL((k_reproject_ps<NG, WAVES, E, S, U>), dim3(grid), dim3(WAVES * 64), 0, st, evA, evB, 0, staged, F.xyz, F.uv, err,   \
                          soft_part, N, F.P, F.W, F.cx, F.cy, clampv, kA, kB, kflags, Nf, F.xyz_stride, F.uv_stride)
    if (ERR && SOFT) { if (UV) DSAC_K2P(true, true, true); else DSAC_K2P(true, true, false); }
    else if (ERR) { if (UV) DSAC_K2P(true, false, true); else DSAC_K2P(true, false, false); }
    else if (SOFT) { if (UV) DSAC_K2P(false, true, true); else DSAC_K2P(false, true, false); }
#undef DSAC_K2P
    return hipGetLastError();
}

template <int NG, int CHW, int WAVES, bool PW, int MINW = 1, bool LO = false, int EX = 0>
static hipError_t launch_reproject_st(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float kA, float kB,
                                      float* soft_part, int* tiles_used, int Nf, int kflags, hipEvent_t evA, hipEvent_t evB, const float* staged_lo = nullptr,
                                      const void* split = nullptr) {
    constexpr int HT = 16 * NG;
    const int tile = WAVES * CHW * 64;
    static_assert(PW || WAVES > 1, "one-wave workgroups write per-wave partial sums");
    const int PT = (F.P + tile - 1) / tile;
    const int NTa = (N + HT - 1) / HT;
    const int grid = (kflags & 32) ? PT * NTa : WAVES == 1 ? ((PT + 31) / 32) * 32 * NTa : ((PT + 7) / 8) * 8 * NTa;
    if (tiles_used) *tiles_used = PW ? PT * WAVES : PT;
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
    const bool G64 = !UV && (F.W & 63) == 0;
#define DSAC_K2S(E, S, U, G)                                                                                                               \
    hipExtLaunchKernelGGL((k_reproject_st<NG, CHW, WAVES, PW, E, S, U, G, MINW, LO, EX>), dim3(grid), dim3(WAVES * 64), 0, st, evA, evB, 0, staged, F.xyz, F.uv, err, \
                          soft_part, N, F.P, F.W, PT, F.cx, F.cy, clampv, kA, kB, kflags, Nf, F.xyz_stride, F.uv_stride, staged_lo, split)
    if (ERR && SOFT) { if (UV) DSAC_K2S(true, true, true, false); else if (G64) DSAC_K2S(true, true, false, true); else DSAC_K2S(true, true, false, false); }
    else if (ERR) { if (UV) DSAC_K2S(true, false, true, false); else if (G64) DSAC_K2S(true, false, false, true); else DSAC_K2S(true, false, false, false); }
    else if (SOFT) { if (UV) DSAC_K2S(false, true, true, false); else if (G64) DSAC_K2S(false, true, false, true); else DSAC_K2S(false, true, false, false); }
#undef DSAC_K2S
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K2, PRECISE form (round 5; dsac_set_option("k2_flags", bit 25)).  The reference projects in double (core/cnn_softam.h:319-362: cv::projectPoints of a
// float point under a double pose) and rounds the residual to float at the end; the fast forms above carry fp32 records into an fp32 matrix-core
// transform, whose four roundings at the magnitude of the partial sums (f * R.X ~ 1e6, z ~ 3e3) are what the suite measures as up to 5.9e-4 px on cells
// close to the camera plane (profiles/r04_final2_parity_margins.txt) -- inside the stated 1e-3, but with a margin of 1.7x, and with a softmax weight in
// a tie of two unrelated hypotheses moving by an estimated 2.3e-4 (stated: 1e-4).  This form evaluates the reference's own arithmetic: the pose records
// in fp64 (Rodrigues of the cv pose in the workgroup's prologue, intrinsics folded in -- no staged fp32 record at all), the rigid transform as three
// chains of three v_fma_f64 per (hypothesis, cell) -- fp64 FMA issues at the rate of an unpacked fp32 FMA on this chip --, the perspective division in
// fp64 through one Newton step on v_rcp_f32's seed, and ONE rounding to float of each image-plane difference; only |d| = sqrt(du^2 + dv^2), the clamp and
// the sigmoid stay in fp32.  Measured residual error against the oracle: see profiles/r05_k2_precise_ab.txt.  Cost: 27 % more issue cycles per pair than
// the matrix-core form (priced), hidden only in part by the store schedule -- a parity mode, not the default.
// Layout: the VALU form's (lane = 4 consecutive cells, hypothesis loop over an LDS tile of records; a wave store is 1 KiB of one error-image row).
// --------------------------------------------------------------------------------------------------
DM_INLINE f2 soft_inlier2_hl(f2 e, float kA, float kAl, float kB, float kBl) {
    const f2 t = pk_fma(splat(kA), e, splat(kB)) + pk_fma(splat(kAl), e, splat(kBl));
    f2 ex;
    ex.x = __builtin_amdgcn_exp2f(t.x);
    ex.y = __builtin_amdgcn_exp2f(t.y);
    const f2 d = ex + splat(1.0f);
    f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return pk_fma(pk_fma(-d, r, splat(1.0f)), r, r);
}

template <int HT, bool ERR, bool SOFT, bool UV, bool DIAG>
__global__ __launch_bounds__(K2_THREADS) void k_reproject_prec(const double* __restrict__ poses, const float* __restrict__ xyz, const float* __restrict__ uv,
                                                               float* __restrict__ err, float* __restrict__ soft_part, int N, int P, int W, int PT,
                                                               float fx, float fy, float cx, float cy, float clampv, float kA, float kB, int Nf,
                                                               long long xyz_stride, long long uv_stride, float kAl, float kBl, int rec32, int diag_arg) {
    const int diag = DIAG ? diag_arg : 0;  // the diagnostic switches only exist in the DIAG instantiation: the parity mode itself pays nothing for them
    const int b = blockIdx.x;
    const int q = b >> 3, PTG = (PT + 7) >> 3;
    const int ht = q / PTG, pt = (q % PTG) * 8 + (b & 7);  // XCD-aware, pixel tiles innermost
    if (pt >= PT) return;
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x;
    {
        const int frame = h0 / Nf;
        xyz += (long long)frame * xyz_stride;
        if (UV) uv += (long long)frame * uv_stride;
    }
    __shared__ __attribute__((aligned(16))) double s_rec[HT * 12];
    __shared__ double s_red[SOFT ? (K2_THREADS / 64) * HT : 1];
    if (tid < nh) {  // record of hypothesis h0 + tid: [fx R0 | fx t0 ; fy R1 | fy t1 ; R2 | t2] in double
        const double* ps = poses + (size_t)(h0 + tid) * 6;
        double r[3] = {ps[0], ps[1], ps[2]}, R[9];
        dm::rodrigues_v2m<false>(r, R, nullptr);
        double* o = s_rec + tid * 12;
        const double dfx = fx, dfy = fy;
        o[0] = dfx * R[0]; o[1] = dfx * R[1]; o[2] = dfx * R[2]; o[3] = dfx * ps[3];
        o[4] = dfy * R[3]; o[5] = dfy * R[4]; o[6] = dfy * R[5]; o[7] = dfy * ps[4];
        o[8] = R[6]; o[9] = R[7]; o[10] = R[8]; o[11] = ps[5];
        // diagnostic (k2_flags bit 26): the records rounded to float as the fast forms stage them, everything else exact -- isolates what the fp32 RECORD alone
        // costs (tests/test_gpu_k2_precise.py: it is the bulk of the fast form's score error, a systematic shift of all of a hypothesis' projections)
        if (rec32) {
#pragma unroll
            for (int k = 0; k < 12; k++) o[k] = (double)(float)o[k];
        }
    }
    const int p0 = (pt * K2_THREADS + tid) * 4;
    const bool valid = p0 < P;  // P % 4 == 0
    double X[4], Y[4], Z[4], pu[4], pv[4];
    if (valid) {
        const f4* src = reinterpret_cast<const f4*>(xyz + (size_t)p0 * 3);
        const f4 a = src[0], bb = src[1], c = src[2];
        X[0] = a.x; Y[0] = a.y; Z[0] = a.z;
        X[1] = a.w; Y[1] = bb.x; Z[1] = bb.y;
        X[2] = bb.z; Y[2] = bb.w; Z[2] = c.x;
        X[3] = c.y; Y[3] = c.z; Z[3] = c.w;
        if (UV) {
            const f4* su = reinterpret_cast<const f4*>(uv + (size_t)p0 * 2);
            const f4 u0 = su[0], u1 = su[1];
            pu[0] = (double)u0.x - (double)cx; pv[0] = (double)u0.y - (double)cy; pu[1] = (double)u0.z - (double)cx; pv[1] = (double)u0.w - (double)cy;
            pu[2] = (double)u1.x - (double)cx; pv[2] = (double)u1.y - (double)cy; pu[3] = (double)u1.z - (double)cx; pv[3] = (double)u1.w - (double)cy;
        } else {
            const int y = p0 / W, x = p0 - y * W;  // 4 | W for the implicit grid of this form: the four cells share a row
#pragma unroll
            for (int k = 0; k < 4; k++) { pu[k] = (double)(x + k) - (double)cx; pv[k] = (double)y - (double)cy; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { X[k] = Y[k] = 0.0; Z[k] = 1.0; pu[k] = pv[k] = 0.0; }
    }
    __syncthreads();
    float* erow = ERR ? err + (size_t)h0 * P + p0 : nullptr;
    const int wave = tid >> 6, lane = tid & 63;
    for (int h = 0; h < nh; h++) {
        const double* sp = s_rec + h * 12;
        const double a0 = sp[0], a1 = sp[1], a2 = sp[2], a3 = sp[3], b0 = sp[4], b1 = sp[5], b2 = sp[6], b3 = sp[7], c0 = sp[8], c1 = sp[9], c2 = sp[10], c3 = sp[11];
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double xc = fma(a0, X[k], fma(a1, Y[k], fma(a2, Z[k], a3)));
            const double yc = fma(b0, X[k], fma(b1, Y[k], fma(b2, Z[k], b3)));
            const double zc = fma(c0, X[k], fma(c1, Y[k], fma(c2, Z[k], c3)));
            // 1 / zc in double: v_rcp_f32's seed (1 ulp, 2^-23) and one Newton step (2^-46: a relative 1.4e-14, i.e. 6e-12 px at the image border -- far
            // below the float the difference is rounded to); projectPoints: z = Z ? 1/Z : 1
            float du, dv;
            if (diag & 4) {
                // diagnostic (k2_diag bit 2, round 6): the exact camera-frame point rounded ONCE to float, then the fast forms' fp32 tail -- what a form
                // with an exact transform in front of hp_chunk's arithmetic would compute (bit 3: one Newton step on v_rcp_f32's result)
                const float xf = (float)xc, yf = (float)yc, zf = (float)zc;
                float izf = __builtin_amdgcn_rcpf(zf);
                if (diag & 8) izf = fmaf(fmaf(-zf, izf, 1.0f), izf, izf);
                izf = (zf == 0.0f) ? 1.0f : izf;
                du = fmaf(-xf, izf, (float)pu[k]);
                dv = fmaf(-yf, izf, (float)pv[k]);
            } else {
                double xq = xc, yq = yc, zq = zc;
                if (diag & 2) { xq = (double)(float)xc; yq = (double)(float)yc; zq = (double)(float)zc; }  // bit 1: the point rounded to float, the tail exact
                double iz = (double)__builtin_amdgcn_rcpf((float)zq);
                iz = fma(fma(-zq, iz, 1.0), iz, iz);
                iz = (zq == 0.0) ? 1.0 : iz;
                du = (float)fma(-xq, iz, pu[k]);  // cell position minus projection: one rounding to float, like the reference's Point2f difference
                dv = (float)fma(-yq, iz, pv[k]);
            }
            e[k] = fminf(__builtin_amdgcn_sqrtf(fmaf(dv, dv, du * du)), clampv);
        }
        if (ERR && valid) __builtin_nontemporal_store(f4{e[0], e[1], e[2], e[3]}, reinterpret_cast<f4*>(erow + (size_t)h * P));
        if (SOFT) {
            // the sums in double from the lane's four sigmoids on: an fp32 tile sum (<= 1024) is rounded to 6e-5, and 300 such tiles put ~1e-3 on a score of
            // ~2e4 -- 2e-4 on a softmax weight in a tie, twice the stated 1e-4
            // ... and the sigmoid without the systematic part of its fp32 error: the exponent from (high, low) pairs of the two constants (a rounded kA / kB shifts
            // every cell's exponent the same way: ~1e-3 on a score), the reciprocal polished by one Newton step
            // k2_diag bit 4: the sigmoid's constants as single floats; bit 5: its reciprocal unpolished; bit 6: the wave's sum in fp32 (the fast forms' sums)
            const float kAl_ = (diag & 16) ? 0.f : kAl, kBl_ = (diag & 16) ? 0.f : kBl;
            f2 s2;
            if (diag & 32) s2 = soft_inlier2(f2{e[0], e[1]}, kA, kB) + soft_inlier2(f2{e[2], e[3]}, kA, kB);
            else s2 = soft_inlier2_hl(f2{e[0], e[1]}, kA, kAl_, kB, kBl_) + soft_inlier2_hl(f2{e[2], e[3]}, kA, kAl_, kB, kBl_);
            double s;
            if (diag & 64) {
                float sf = valid ? s2.x + s2.y : 0.f;
                sf = wave_sum(sf);
                s = (double)sf;
            } else {
                s = valid ? (double)s2.x + (double)s2.y : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            }
            if (lane == 0) s_red[wave * HT + h] = s;
        }
    }
    if (SOFT) {
        __syncthreads();
        if (tid < nh) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < K2_THREADS / 64; w++) s += s_red[w * HT + tid];
            // the partial rows are floats: the tile's sum travels as a (high, low) pair of rows, which the fp64 reduction adds back together
            const float hi = (float)s;
            soft_part[(size_t)pt * N + h0 + tid] = hi;
            soft_part[(size_t)(PT + pt) * N + h0 + tid] = (float)(s - (double)hi);
        }
    }
}

template <int HT>
static hipError_t launch_reproject_prec(hipStream_t st, int N, const double* poses, const FrameDev& F, float clampv, float* err, double kAd, double kBd,
                                        float* soft_part, int* tiles_used, int Nf, hipEvent_t evA, hipEvent_t evB, int rec32, int diag) {
    const float kA = (float)kAd, kB = (float)kBd, kAl = (float)(kAd - (double)kA), kBl = (float)(kBd - (double)kB);
    const int PT = (F.P + K2_THREADS * 4 - 1) / (K2_THREADS * 4);
    const int NTa = (N + HT - 1) / HT;
    const int grid = ((PT + 7) / 8) * 8 * NTa;
    if (tiles_used) *tiles_used = 2 * PT;  // a (high, low) pair of partial rows per pixel tile; 2 ceil(P / 1024) <= reproject_num_pixel_tiles(P)
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
#define DSAC_K2P(E, S, U)                                                                                                                       \
    do { if (diag) hipExtLaunchKernelGGL((k_reproject_prec<HT, E, S, U, true>), dim3(grid), dim3(K2_THREADS), 0, st, evA, evB, 0, poses, F.xyz, F.uv, err, soft_part, N, F.P, \
                          F.W, PT, F.fx, F.fy, F.cx, F.cy, clampv, kA, kB, Nf, F.xyz_stride, F.uv_stride, kAl, kBl, rec32, diag); \
         else hipExtLaunchKernelGGL((k_reproject_prec<HT, E, S, U, false>), dim3(grid), dim3(K2_THREADS), 0, st, evA, evB, 0, poses, F.xyz, F.uv, err, soft_part, N, F.P, \
                          F.W, PT, F.fx, F.fy, F.cx, F.cy, clampv, kA, kB, Nf, F.xyz_stride, F.uv_stride, kAl, kBl, rec32, 0); } while (0)
    if (ERR && SOFT) { if (UV) DSAC_K2P(true, true, true); else DSAC_K2P(true, true, false); }
    else if (ERR) { if (UV) DSAC_K2P(true, false, true); else DSAC_K2P(true, false, false); }
    else if (SOFT) { if (UV) DSAC_K2P(false, true, true); else DSAC_K2P(false, true, false); }
#undef DSAC_K2P
    return hipGetLastError();
}

bool reproject_variant_known(int v) {
    return v == -1 || (v >= 0 && v <= 3) || (v >= 10 && v <= 13) || (v >= 20 && v <= 27) || (v >= 40 && v <= 62) || (v >= 65 && v <= 77) || (v >= 80 && v <= 85) || v == 89 || (v >= 93 && v <= 95);
}

// the largest count of partial-sum rows over all forms: one row per 64-pixel wave chunk, and the per-wave-sum forms with several waves per workgroup write
// PT * WAVES rows (their idle waves of the last workgroup write zero rows): up to WAVES - 1 <= 15 rows beyond ceil(P / 64)
int reproject_num_pixel_tiles(int P) { return (P + 63) / 64 + 15; }

template <int PX, int HT, bool SPOSE>
static hipError_t launch_reproject(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float kA, float kB,
                                   float* soft_part, int Nf, bool pixel_minor, int kflags, hipEvent_t evA, hipEvent_t evB) {
    const int tile = K2_THREADS * PX;
    const int PT = (F.P + tile - 1) / tile;
    const int NTa = (N + HT - 1) / HT;
    const int grid = ((PT + 7) / 8) * 8 * NTa;
    const int NT = pixel_minor ? -NTa : NTa;
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
#define DSAC_K2(E, S, U)                                                                                                              \
    hipExtLaunchKernelGGL((k_reproject<PX, HT, E, S, U, SPOSE>), dim3(grid), dim3(K2_THREADS), 0, st, evA, evB, 0, staged, F.xyz, F.uv, err, soft_part, \
                          N, F.P, F.W, PT, NT, F.cx, F.cy, clampv, kA, kB, kflags, Nf, F.xyz_stride, F.uv_stride)
    if (ERR && SOFT) { if (UV) DSAC_K2(true, true, true); else DSAC_K2(true, true, false); }
    else if (ERR) { if (UV) DSAC_K2(true, false, true); else DSAC_K2(true, false, false); }
    else if (SOFT) { if (UV) DSAC_K2(false, true, true); else DSAC_K2(false, true, false); }
#undef DSAC_K2
    return hipGetLastError();
}

// variant: -1 = auto (default) ; VALU forms: 0 = LDS-staged poses, HT = 32 ; 1 = scalar-load poses (SGPR operands), HT = 32 ; 2 = HT 16 ;
//          3 = HT 64 ; 10..13 = 1, 2, 4, 8 hypothesis rows per workgroup ; matrix-core forms <HT, 64-pixel chunks per wave>: 20 = <64,2>,
//          21 = <64,4>, 22 = <128,2>, 23 = <32,2>, 24 = <64,1>, 25 = <128,4>, 26 = <32,4>, 27 = <128,1> ; streaming small-tile forms
//          <16-hypothesis groups, chunks per wave, waves per workgroup>: 40 = <1,1,4>, 41 = <2,1,4>, 42 = <4,1,4>, 43 = <2,2,4> (partial sums
//          through LDS, one row per workgroup) ; per-wave partial sums, no barrier: 44-46 = <1|2|4,4,1>, 47-49 = <1|2|4,2,1>, 50-52 = <1|2|4,1,1>,
//          53 = <2,1,4>, 54 = <2,2,4>, 55 = <4,1,4>, 56 = <2,4,4>, 57 = <4,2,4>.  Unknown values are an error.
hipError_t reproject(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float tau, float beta,
                     float* soft_part, const K2Opts& opts, int* tiles_used, int Nf) {
    if (tiles_used) *tiles_used = 0;
    if (N <= 0 || F.P <= 0 || (!err && !soft_part)) return hipSuccess;
    if (Nf <= 0 || F.frames <= 1) Nf = N > 0 ? ((N + 127) / 128) * 128 : 128;  // one frame: every tile maps to frame 0
    else if (Nf % K2_NF_MULTIPLE != 0) return hipErrorInvalidValue;           // tiles (<= 128 hypotheses) must not straddle frames
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(F.uv) & 15) == 0);
    if (tiles_used) *tiles_used = vec ? (F.P + K2_THREADS * 4 - 1) / (K2_THREADS * 4) : (F.P + K2_THREADS - 1) / K2_THREADS;
    // the streaming / persistent forms (variant >= 40) know two block orders: XCD-aware pixel-minor (k2_order 1) and plain pixel-minor (k2_order 0
    // or k2_flags bit 5)
    int kf = opts.flags | ((opts.variant >= 40 && !opts.pixel_minor) ? 32 : 0);
    // timing events (dsac_profile_enable): attached to the kernel's own dispatch (hipExtLaunchKernelGGL) instead of two event records on the
    // stream, which cost ~7 us of bubble each (one 640x480 frame: 100 us per step with them, 86 without)
    hipEvent_t evA = opts.ev_start, evB = opts.ev_stop;
    // An arithmetic form that was ASKED for (k2_flags bits 25 / 27 / 28) and cannot run on this map -- its kernels read 16-byte vectors: H*W % 4, aligned buffers
    // and, on the implicit grid, four cells of a lane in one row; the exact form also needs a focal length <= 2^10 -- is an error, not a silent fp32 launch
    // (a parity run must not measure the fast form unknowingly).  The auto policy's exact form simply falls back.
    if (opts.poses64 && (((opts.flags & K2_FLAG_PRECISE) && !(vec && (F.uv || F.W % 4 == 0))) || ((opts.flags & K2_FLAG_RECLO) && !vec) ||
                         ((opts.flags & K2_FLAG_EXACT) && !(vec && opts.split))))
        return hipErrorNotSupported;
    // the precise form (k2_flags bit 25): needs the cv poses themselves, 16-byte vectors and -- on the implicit grid -- four cells of a lane in one row
    if ((opts.flags & K2_FLAG_PRECISE) && opts.poses64 && vec && (F.uv || F.W % 4 == 0)) {
        const double nhp = (double)N * (double)F.P;
        const double kAd = (double)beta * 1.4426950408889634, kBd = -(double)beta * (double)tau * 1.4426950408889634;
        // hypothesis tile by size as the VALU forms: 32 for big launches, 16 for a single small frame (more workgroups)
        const int rec32 = (opts.flags >> 26) & 1;
        return nhp > 2.0e8 ? launch_reproject_prec<32>(st, N, opts.poses64, F, clampv, err, kAd, kBd, soft_part, tiles_used, Nf, evA, evB, rec32, opts.diag)
                           : launch_reproject_prec<16>(st, N, opts.poses64, F, clampv, err, kAd, kBd, soft_part, tiles_used, Nf, evA, evB, rec32, opts.diag);
    }
    // records in two pieces (k2_flags bit 27): the one-wave streaming form <64 hypotheses, 256 pixels> with the fp16 correction instructions, whatever the size
    if ((opts.flags & K2_FLAG_RECLO) && opts.staged_lo && vec && (opts.variant < 0 || (opts.variant >= 80 && opts.variant <= 83))) {
#define DSAC_LO(NG_, CH_, MW_) launch_reproject_st<NG_, CH_, 1, true, MW_, true>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf | 32, evA, evB, opts.staged_lo)
        switch (opts.variant) {  // k2_variant 80..83 with the flag: the register / occupancy trades of this form (A/B); -1: the measured one
            case 80: return DSAC_LO(4, 4, 3);   // <64 hypotheses, 256 pixels>, 3 waves per SIMD (92 B of scratch)
            case 82: return DSAC_LO(4, 2, 4);   // <64, 128>, 4 waves per SIMD
            case 83: return DSAC_LO(2, 4, 3);   // <32, 256>, 3 waves per SIMD
            case 81: default: return DSAC_LO(4, 4, 2);  // <64, 256>, 2 waves per SIMD, no scratch: the fastest of the four (profiles/r05_k2_reclo_ab.txt)
        }
#undef DSAC_LO
    }
    // the exact-transform form (k2_flags bit 28, round 6): the one-wave streaming forms with the split records; k2_variant 84 / 89 / 93 = its tile / occupancy trades
    if (k2_wants_exact(opts) && opts.split && vec && (opts.variant < 0 || (opts.variant == 84 || opts.variant == 85 || opts.variant == 89 || (opts.variant >= 93 && opts.variant <= 95)))) {
#define DSAC_EX(NG_, CH_, MW_, F_) launch_reproject_st<NG_, CH_, 1, true, MW_, false, F_>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf | 32, evA, evB, nullptr, opts.split)
        switch (opts.variant) {
            // (85 .. 88: <64, 128 / 256> with a scheduling fence per m at three waves per SIMD, one / two / four waves per workgroup -- 44-480 B of scratch, 1.08-1.7 ms:
            //  measured and removed, profiles/r06_k2_exact_ab.txt)
            case 85: return DSAC_EX(4, 4, 2, 2);      // 84 with the one-transcendental tail (A/B)
            case 94: return DSAC_EX(4, 1, 3, 2);      // 93 with the one-transcendental tail (A/B)
            case 89: return DSAC_EX(2, 4, 2, 1);      // <32 hypotheses, 256 pixels>, 2 waves per SIMD
            case 93: return DSAC_EX(4, 1, 3, 1);      // <64, 64>, one-wave workgroups, 168 registers: 3 waves per SIMD
            // (90 .. 92: <64, 4 waves x 64>, <64, 2 x 128> at two / three waves per SIMD: 1 098-1 206 us at the bench shape, 67.5-77 us for one frame -- measured
            //  and removed, profiles/r06_k2_exact_tiles.txt)
            case 95: return DSAC_EX(4, 1, 4, 2);      // 94 at four waves per SIMD (the one-transcendental tail needs 128 registers there)
            case -1:  // auto: the one-transcendental tail (968-987 against 1 002-1 020 us at the bench shape, the same or smaller errors: profiles/r06_k2_rsq_ab.txt,
                      // r06_k2_rsq_parity.txt) on the <64, 256> tile at every size.  For one frame of 256 hypotheses the <64, 64> tile is 0.8 us faster by itself
                      // (62.6 against 63.4) but leaves four times the partial-sum rows: k_reduce_soft 27.6 against 8.9 us (profiles/r06_one_image_trace.txt)
                // small maps keep the small tile: a 40 x 40 map is 7 tiles of 256 pixels (one image per call 116 -> 127 us with them), and its partial sums are 25 rows
                if (F.P <= 16384) return DSAC_EX(4, 1, 3, 2);
                return DSAC_EX(4, 4, 2, 2);
            case 84: default: return DSAC_EX(4, 4, 2, 1);  // <64, 256>, 2 waves per SIMD, reciprocal + Newton + square root (<64, 256> at three waves per SIMD spills: 1.7 ms, profiles/r06_k2_exact_ab.txt)
        }
#undef DSAC_EX
    }
    if (!vec) return launch_reproject<1, 32, false>(st, N, staged, F, clampv, err, kA, kB, soft_part, Nf, opts.pixel_minor, kf, evA, evB);
    const bool pm = opts.variant < 0 ? true : opts.pixel_minor;  // the auto policy's forms were all measured with pixel tiles innermost
#define DSAC_VA(PX_, HT_, SP_) launch_reproject<PX_, HT_, SP_>(st, N, staged, F, clampv, err, kA, kB, soft_part, Nf, pm, kf, evA, evB)
#define DSAC_HP(HT_, CH_) launch_reproject_hp<HT_, CH_>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, pm, kf, evA, evB)
#define DSAC_ST(NG_, CH_, WV_, PW_) launch_reproject_st<NG_, CH_, WV_, PW_>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB)
    int variant = opts.variant;
    if (variant < 0) {
        // auto (measured on MI355X, profiles/r02_k2_variants.txt).  With the fused soft-inlier sums the kernel is VALU-limited: matrix-core
        // form; 1024 pixels per workgroup for big launches (> 1.5 GB of error images: N = 4096 or a batch of frames: 443 vs 460-508 us for
        // the 8-frame batch), 256 for a single frame of 256 hypotheses.  Error images only: store-limited, the VALU kernel with pixel
        // tiles innermost has the best store stream (N = 256: 56.6 vs 58.9 us; N = 4096: equal).
        // Round 3 (A/B inside bench.py on five boxes, profiles/r03_k2_ab_bench*.txt -- a plain-HIP harness, scripts/micro/k2_lab, ranks the forms
        // differently from the bench's process and is NOT what this policy follows): big launches take the streaming form <32 hypotheses, 256
        // pixels> per one-wave workgroup (no LDS, no barrier, operands straight from the staged records, per-wave partial sums) in PLAIN
        // pixel-minor block order: 842-870 us against 904-950 for round 2's <64,4> (0.73-0.75 of the HBM spec instead of 0.67-0.70); one frame
        // of 256 hypotheses takes <64 hypotheses, 4 waves x 64 pixels>: 56.7-58.0 against 60.2-60.8 us.
        // Late round 3: <64 hypotheses, 256 pixels> per one-wave workgroup at >= 4 waves per SIMD (form 58) instead of <32, 256> (form 45): 1 % ahead in
        // every A/B on six boxes (profiles/r03_k2_ab_58_45.txt: 861-867 against 872-875 us alternating on one box) and half the partial-sum rows
        // (with sampled pixel positions, or a map width that is not a multiple of 64, the kernel keeps the positions of every chunk in vector
        // registers: form 58's 128-register build would spill 108-116 bytes per lane there, form 45 needs no scratch)
        // By size (A/B inside bench.py at 2 ... 16 frames of 256 hypotheses per launch, profiles/r03_k2_ab_mid.txt): up to 1 GB of error images form 42
        // (2 frames: 111.7 us against 115.1 for form 45 and 121.6 for 58), up to 4.4 GB form 45 (4 frames 220.5 against 224 / 232; 8 frames
        // 424-427 us = 0.75 of the HBM spec against 443 / 460-465; 12 frames 644 = 638-644), above it form 58 (16 frames: 861-867 against 872-875)
        const double bytes = (double)N * (double)F.P * 4.0;
        const bool big = bytes > 1.0e9;
        const bool grid64 = F.uv == nullptr && (F.W & 63) == 0;
        variant = soft_part ? (big ? ((grid64 && bytes > 4.4e9) ? 58 : 45) : 42) : 0;
        if (soft_part && big) kf |= 32;
    }
    switch (variant) {
        case 0: return DSAC_VA(4, 32, false);
        case 1: return DSAC_VA(4, 32, true);
        case 2: return DSAC_VA(4, 16, false);
        case 3: return DSAC_VA(4, 64, false);
        case 10: return DSAC_VA(4, 1, false);
        case 11: return DSAC_VA(4, 2, false);
        case 12: return DSAC_VA(4, 4, false);
        case 13: return DSAC_VA(4, 8, false);
        case 20: return DSAC_HP(64, 2);
        case 21: return DSAC_HP(64, 4);
        case 22: return DSAC_HP(128, 2);
        case 23: return DSAC_HP(32, 2);
        case 24: return DSAC_HP(64, 1);
        case 25: return DSAC_HP(128, 4);
        case 26: return DSAC_HP(32, 4);
        case 27: return DSAC_HP(128, 1);
        // streaming small-tile forms <16-hypothesis groups, 64-pixel chunks per wave, waves per workgroup>
        case 40: return DSAC_ST(1, 1, 4, false);
        case 41: return DSAC_ST(2, 1, 4, false);
        case 42: return DSAC_ST(4, 1, 4, false);
        case 43: return DSAC_ST(2, 2, 4, false);
        case 44: return DSAC_ST(1, 4, 1, true);
        case 45: return DSAC_ST(2, 4, 1, true);
        case 46: return DSAC_ST(4, 4, 1, true);
        case 47: return DSAC_ST(1, 2, 1, true);
        case 48: return DSAC_ST(2, 2, 1, true);
        case 49: return DSAC_ST(4, 2, 1, true);
        case 50: return DSAC_ST(1, 1, 1, true);
        case 51: return DSAC_ST(2, 1, 1, true);
        case 52: return DSAC_ST(4, 1, 1, true);
        case 53: return DSAC_ST(2, 1, 4, true);
        case 54: return DSAC_ST(2, 2, 4, true);
        case 55: return DSAC_ST(4, 1, 4, true);
        case 56: return DSAC_ST(2, 4, 4, true);
        case 57: return DSAC_ST(4, 2, 4, true);
        case 58: case 80: case 81: case 82: case 83: case 84: case 85: case 89: case 93: case 94: case 95:  // 84 / 89 / 93: the exact-transform forms when k2_flags bit 28 is set (above); 80..83: the two-piece-record forms when k2_flags bit 27 is set (above); without the flag the plain form
            return launch_reproject_st<4, 4, 1, true, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);  // <4,4,1>, >= 4 waves per SIMD
        case 59: return launch_reproject_st<2, 4, 1, true, 5>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);  // <2,4,1>, >= 5 waves per SIMD
        case 65: return launch_reproject_st<4, 4, 4, true, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);  // <4,4,4> per-wave sums
        case 66: return launch_reproject_st<4, 4, 2, true, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);  // <4,4,2>
        case 67: return DSAC_ST(1, 4, 4, true);
        case 68: return DSAC_ST(2, 4, 4, false);
        case 69: return DSAC_ST(2, 4, 8, true);
        case 70: return launch_reproject_st<2, 4, 4, true, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);
        case 71: return DSAC_ST(1, 4, 8, true);
        case 72: return DSAC_ST(2, 4, 16, true);
        case 73: return DSAC_ST(2, 8, 1, true);
        case 74: return DSAC_ST(1, 8, 1, true);
        case 75: return DSAC_ST(2, 6, 1, true);
        case 76: return DSAC_ST(2, 3, 1, true);
        case 77: return DSAC_ST(1, 6, 1, true);
        // persistent pipelined forms <16-hypothesis groups, waves per workgroup>; k2_flags bits 8..15 = workgroups per CU
        case 60: return launch_reproject_ps<1, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);
        case 61: return launch_reproject_ps<2, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);
        case 62: return launch_reproject_ps<4, 4>(st, N, staged, F, clampv, err, kA, kB, soft_part, tiles_used, Nf, kf, evA, evB);
        default: return hipErrorInvalidValue;
    }
#undef DSAC_VA
#undef DSAC_HP
#undef DSAC_ST
}

// --------------------------------------------------------------------------------------------------
// A workgroup of 16 waves per 64 hypotheses: lane = hypothesis (a wave reads 256 contiguous bytes of a partial row), wave w sums the rows
// t = w, w + 16, ... in fp64, then a fixed-order sum over the 16 waves -> deterministic.  (Rounds 1-2 ran one wave per hypothesis with the
// lanes striding over the rows: every lane touched its own 64-byte sector, fine for 300 rows, not for the 1200 of the one-wave tiles.)
constexpr int KR_WAVES = 16;
__global__ __launch_bounds__(KR_WAVES * 64) void k_reduce_soft(int N, int tiles, const float* __restrict__ part, double* __restrict__ soft) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ double s_acc[KR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    double s = 0;
    if (h < N) {
#pragma unroll 16  // sixteen row reads in flight per wave (the adds stay in row order): 1 200 rows are 75 per wave, a chain of latencies at four
        for (int t = wave; t < tiles; t += KR_WAVES) s += (double)part[(size_t)t * N + h];
    }
    s_acc[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && h < N) {
        double tot = 0;
#pragma unroll
        for (int w = 0; w < KR_WAVES; w++) tot += s_acc[w][lane];
        soft[h] = tot;
    }
}

hipError_t reduce_soft(hipStream_t st, int N, int tiles, const float* soft_part, double* soft) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_soft, dim3((N + 63) / 64), dim3(KR_WAVES * 64), 0, st, N, tiles, soft_part, soft);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K3: softmax (max-subtracted), Shannon entropy in bits, weighted pose average.  One workgroup, fp64,
// fixed reduction tree -> deterministic.  Replaces core/cnn_softam.h:535-553, :80-88, :1082-1094.
// --------------------------------------------------------------------------------------------------
constexpr int K3_THREADS = 256;

DM_INLINE double wave_allmax_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
DM_INLINE double wave_allsum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Two block-wide reductions in total: the maximum, then {sum e, sum e*log2 e, sum e*pose[0..5]} together.
// With S = sum e:  w_i = e_i / S,  entropy = log2 S - (sum e log2 e)/S,  avg = (sum e pose)/S.
__global__ __launch_bounds__(K3_THREADS) void k_softmax(int N, const double* __restrict__ scores, double scale, double* __restrict__ w,
                                                        double* __restrict__ entropy, const double* __restrict__ poses,
                                                        double* __restrict__ avg6) {
    __builtin_amdgcn_s_setprio(3);  // tiny latency-bound kernel, usually overlapped with a bandwidth-bound one
    {   // frame batch: workgroup f owns the N scores of frame f
        const size_t f = blockIdx.x;
        scores += f * N; w += f * N;
        if (entropy) entropy += f;
        if (poses) poses += f * N * 6;
        if (avg6) avg6 += f * 6;
    }
    __shared__ double s_m[K3_THREADS / 64];
    __shared__ double s_acc[K3_THREADS / 64][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool want_avg = poses && avg6;
    double m = -1.7976931348623157e308;
    for (int i = tid; i < N; i += K3_THREADS) m = fmax(m, scale * scores[i]);
    m = wave_allmax_d(m);
    if (lane == 0) s_m[wave] = m;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K3_THREADS / 64; k++) m = fmax(m, s_m[k]);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < N; i += K3_THREADS) {
        const double x = scale * scores[i] - m;
        const double e = exp(x);
        w[i] = e;
        acc[0] += e;
        acc[1] += e * (x * 1.4426950408889634);  // e * log2(e); exactly 0 when e underflows, like the reference's w > 0 test
        if (want_avg) {
#pragma unroll
            for (int k = 0; k < 6; k++) acc[2 + k] += e * poses[6 * (size_t)i + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = wave_allsum_d(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) s_acc[wave][k] = acc[k];
    }
    __syncthreads();
    double tot[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        tot[k] = 0;
#pragma unroll
        for (int q = 0; q < K3_THREADS / 64; q++) tot[k] += s_acc[q][k];
    }
    const double S = tot[0];
    for (int i = tid; i < N; i += K3_THREADS) w[i] = w[i] / S;
    if (tid == 0) {
        if (entropy) *entropy = log2(S) - tot[1] / S;
        if (want_avg) {
#pragma unroll
            for (int k = 0; k < 6; k++) avg6[k] = tot[2 + k] / S;
        }
    }
}

hipError_t softmax(hipStream_t st, int N, const double* scores, double scale, double* w, double* entropy, const double* poses, double* avg6,
                   int frames) {
    if (N <= 0 || frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_softmax, dim3(frames), dim3(K3_THREADS), 0, st, N, scores, scale, w, entropy, poses, avg6);
    return hipGetLastError();
}

}  // namespace dk

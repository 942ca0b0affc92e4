// k_backward.hip -- K4: backward of the hypothesis scores to per-pixel scene-coordinate gradients.
//
// Replaces dScore part (iii) (core/cnn_softam.h:609-645: dProjectdObj :404-453, dProjectdHyp :464-528 per
// (hypothesis, pixel)) and the sum over hypotheses at core/train_ransac_softam.cpp:382-383.
//
// Restructuring that makes it a streaming kernel.  In the jp convention E = R'X + t',
//   px = -f E.x/E.z + cx,  py = f E.y/E.z + cy,  err = |(u,v) - (px,py)|,  a = -(u - px, v - py)/(err + 1e-8),
// define the per-(h,p) 3-vector  c = ( -a0 f/E.z ,  a1 f/E.z ,  (a0 E.x - a1 E.y) f/E.z^2 ).  Then
//   dProjectdObj = R'^T c                       (1 x 3, per pixel)
//   dProjectdHyp = [ (c (x) X) : dR'/drod , c ] (1 x 6)  -- the 2x9 dPdR of the reference is rank-structured,
// and d R'/d rod depends only on the hypothesis, so the pixel loop only has to accumulate
//   grad[p]  += w R'^T c         (register accumulation over the hypothesis loop)
//   G12[h]   += w [c (x) X, c]   (12 sums over pixels per hypothesis)
// with w = d_err[h][p]; the 9x3 Rodrigues derivative and the 6x12 dPNP are applied once per hypothesis in
// the finish kernel.  The reference calls cv::Rodrigues twice per pixel for that (core/cnn_softam.h:507-508).
//
// The main pass reads d_err (N x P f32, streaming) once: 4 B per (h,p), HBM-read bound on paper and close
// to the fp32 VALU ridge in practice (~55 VALU ops per pair).  All reductions are two-stage and
// deterministic except the final fp64 scatter to the 4 support pixels (atomics).
#include "kernels.h"
#include "dmath.h"

namespace dk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int K4_THREADS = 256;
constexpr int K4_PXG = 2;     // 8 pixels per lane on the vector path
constexpr int K4_HT_MAX = 128;  // LDS capacity for the hypothesis records of a tile; the tile size itself is chosen per launch
constexpr int BWD_REC = BWD_STRIDE;
// Record of one hypothesis, six float4 read by K4 with ds_read_b128 and used as they are (jp convention: E = R' X + t'):
//   [0] ( R'00, -R'10,  R'01, -R'11)   column pairs for the packed chain (E.x, -E.y) = c0 X + c1 Y + c2 Z + c3
//   [1] ( R'02, -R'12,  t'0,  -t'1 )
//   [2] ( R'20,  R'21,  R'22,  t'2 )   E.z
//   [3] ( R'00,  R'01, -R'10, -R'11)   gx.xy += (R'0.xy) C0 + (-R'1.xy)(-C1) + (-R'2.xy)(-C2)
//   [4] (-R'20, -R'21,  R'02, -R'12)
//   [5] (-R'22,  0, 0, 0)              gx.z



// --------------------------------------------------------------------------------------------------
// per hypothesis: jp pose (cv2our), its float record, and dR'/drod (3x9) for the finish kernel
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backward_prep(int N, const double* __restrict__ poses, float* __restrict__ rec,
                                                      double* __restrict__ dRdH) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    double cv6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cv6[k] = poses[(size_t)h * 6 + k];
    double R[9], t[3];
    dm::cv2our(cv6, R, t);
    float* o = rec + (size_t)h * BWD_REC;
    const float r00 = (float)R[0], r01 = (float)R[1], r02 = (float)R[2], r10 = (float)R[3], r11 = (float)R[4], r12 = (float)R[5];
    const float r20 = (float)R[6], r21 = (float)R[7], r22 = (float)R[8], t0 = (float)t[0], t1 = (float)t[1], t2 = (float)t[2];
    o[0] = r00; o[1] = -r10; o[2] = r01; o[3] = -r11;
    o[4] = r02; o[5] = -r12; o[6] = t0; o[7] = -t1;
    o[8] = r20; o[9] = r21; o[10] = r22; o[11] = t2;
    o[12] = r00; o[13] = r01; o[14] = -r10; o[15] = -r11;
    o[16] = -r20; o[17] = -r21; o[18] = r02; o[19] = -r12;
    o[20] = -r22; o[21] = 0.f; o[22] = 0.f; o[23] = 0.f;
    // rod = Rodrigues(R'), dRdH = d Rodrigues(rod) / d rod   (core/cnn_softam.h:505-509)
    double rod[3], Rre[9], J[27];
    dm::rodrigues_m2v(R, rod);
    dm::rodrigues_v2m<true>(rod, Rre, J);
#pragma unroll
    for (int k = 0; k < 27; k++) dRdH[(size_t)h * 27 + k] = J[k];
}

hipError_t backward_prep(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged_bwd, double* dRdH) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_backward_prep, dim3((N + 63) / 64), dim3(64), 0, st, N, poses, staged_bwd, dRdH);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// Transpose-reduce of 12 per-lane values over the 64 lanes of a wave (26 + 9 VALU ops instead of 12 x 6 DPP steps
// with their adds): every step halves both the number of live values per lane and the number of lanes left to sum.
//   v_permlane32_swap / v_permlane16_swap (gfx950): swap(x, y) exchanges the upper half (odd 16-rows) of x with the
//   lower half (even rows) of y, so x + y afterwards holds "value x summed over both halves" in the lower lanes and
//   "value y summed" in the upper ones.  Inside a 16-row the same idea runs on quad_perm DPP moves with per-lane selects,
//   then two row rotations finish the sum over lane bits 2 and 3.
// Lane L returns the wave total of value index
//   6*(L>>5) + 3*((L>>4)&1) + {0 if L&3==0, 1 if L&3==2, 2 if L&3==1}      (lanes with L&3 == 3 hold nothing)
// (scripts/micro/wave_sum12.hip checks this mapping on the device.)
DM_INLINE void lane_swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
DM_INLINE void lane_swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
template <int CTRL, int BANK>
DM_INLINE float dpp_merge(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, BANK, false));
}
DM_INLINE float wave_sum12(float (&a)[12], int lane) {
#pragma unroll
    for (int k = 0; k < 6; k++) { lane_swap32(a[k], a[6 + k]); a[k] += a[6 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { lane_swap16(a[k], a[3 + k]); a[k] += a[3 + k]; }
    const bool odd = lane & 1, b1 = lane & 2;
    // lane ^ 1: even lanes keep (a0, a1), odd lanes keep (a2, -); each lane sends what its partner keeps
    const float s0 = (odd ? a[2] : a[0]) + dpp_merge<0xB1, 0xf>(0.f, odd ? a[0] : a[2]);
    const float s1 = (odd ? 0.f : a[1]) + dpp_merge<0xB1, 0xf>(0.f, odd ? a[1] : 0.f);
    // lane ^ 2: bit1 = 0 keeps s0, bit1 = 1 keeps s1
    float v = (b1 ? s1 : s0) + dpp_merge<0x4E, 0xf>(0.f, b1 ? s0 : s1);
    v += dpp_merge<0x124, 0xf>(0.f, v);  // row_ror:4
    v += dpp_merge<0x128, 0xf>(0.f, v);  // row_ror:8
    return v;
}

// K4 main pass.  A lane owns PXG groups of 4 consecutive pixels (group j at tile0 + j*1024 + 4*tid), so every
// d_err / xyz / grad access is a coalesced dwordx4 and the per-hypothesis 12-value wave reduction is amortised
// over 4*PXG pixels.  No barrier inside the hypothesis loop: every wave writes its own partial row.
//   grad_part : [hyp tile][P*3]                 (register accumulation over the 32 hypotheses of the tile)
//   G12_part  : [pixel tile * 4 + wave][N][12]  (sum over the wave's pixels)
// SOFTMODE: w = g[h] * d soft / d err, soft = sigmoid(beta (tau - min(err, clamp)))
template <int PXG, bool VEC, bool SOFTMODE, bool UV>
__global__ __launch_bounds__(K4_THREADS) void k_score_backward(const float* __restrict__ rec, const float* __restrict__ xyz,
                                                               const float* __restrict__ uv, const float* __restrict__ d_err,
                                                               const double* __restrict__ g, float* __restrict__ grad_part,
                                                               float* __restrict__ G12_part, int N, int P, int W, int PT, int NT,
                                                               float f, float cx, float cy, float clampv, float kA, float kB, float beta, int HT) {
    constexpr int PXL = VEC ? 4 : 1;          // pixels per group
    constexpr int NP = PXG * PXL;             // pixels per lane
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ht = q % NT;
    const int pt = (q / NT) * 8 + (b & 7);
    if (pt >= PT) return;
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    __shared__ __attribute__((aligned(16))) float s_rec[K4_HT_MAX * BWD_REC];
    __shared__ float s_g[K4_HT_MAX];
    for (int i = tid; i < nh * BWD_REC; i += K4_THREADS) s_rec[i] = rec[(size_t)h0 * BWD_REC + i];
    if (SOFTMODE && tid < nh) s_g[tid] = (float)g[h0 + tid];  // nh <= K4_HT_MAX <= K4_THREADS

    const int tile0 = pt * K4_THREADS * NP;
    int pbase[PXG];
    bool valid[PXG];
    // per pixel: (X, Y) and (Z, 1) as register pairs (operands of v_pk_fma_f32), (u - cx, v - cy) as a pair
    f2 xy[NP], zw[NP], ppix[NP];
#pragma unroll
    for (int j = 0; j < PXG; j++) {
        pbase[j] = tile0 + j * K4_THREADS * PXL + tid * PXL;
        valid[j] = pbase[j] < P;  // VEC is only used with P % 4 == 0
        if (VEC) {
            if (valid[j]) {
                const f4* src = reinterpret_cast<const f4*>(xyz + (size_t)pbase[j] * 3);
                const f4 a = src[0], bb = src[1], c = src[2];
                xy[j * 4 + 0] = f2{a.x, a.y};   zw[j * 4 + 0] = f2{a.z, 1.f};
                xy[j * 4 + 1] = f2{a.w, bb.x};  zw[j * 4 + 1] = f2{bb.y, 1.f};
                xy[j * 4 + 2] = f2{bb.z, bb.w}; zw[j * 4 + 2] = f2{c.x, 1.f};
                xy[j * 4 + 3] = f2{c.y, c.z};   zw[j * 4 + 3] = f2{c.w, 1.f};
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) { xy[j * 4 + k] = f2{0.f, 0.f}; zw[j * 4 + k] = f2{0.f, 1.f}; }
            }
        } else {
            if (valid[j]) { xy[j] = f2{xyz[(size_t)pbase[j] * 3], xyz[(size_t)pbase[j] * 3 + 1]}; zw[j] = f2{xyz[(size_t)pbase[j] * 3 + 2], 1.f}; }
            else { xy[j] = f2{0.f, 0.f}; zw[j] = f2{0.f, 1.f}; }
        }
        if (UV) {
#pragma unroll
            for (int k = 0; k < PXL; k++) {
                const int p = pbase[j] + k;
                ppix[j * PXL + k] = valid[j] ? f2{uv[(size_t)p * 2] - cx, uv[(size_t)p * 2 + 1] - cy} : f2{0.f, 0.f};
            }
        } else {
            int y = pbase[j] / W, x = pbase[j] - y * W;  // one division per group, then walk with row wrap
#pragma unroll
            for (int k = 0; k < PXL; k++) {
                ppix[j * PXL + k] = f2{(float)x - cx, (float)y - cy};
                if (++x == W) { x = 0; y++; }
            }
        }
    }
    __syncthreads();

    f2 gxy[NP];
    float gz[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) { gxy[k] = f2{0.f, 0.f}; gz[k] = 0.f; }

    float* gout = G12_part + ((size_t)(pt * (K4_THREADS / 64) + wave) * N + h0) * 12;
    // which of the 12 per-hypothesis sums this lane holds after wave_sum12 (-1: none)
    const int gslot = ((lane & 12) == 0 && (lane & 3) != 3) ? 6 * (lane >> 5) + 3 * ((lane >> 4) & 1) + ((lane & 3) == 0 ? 0 : (lane & 3) == 2 ? 1 : 2) : -1;
    for (int h = 0; h < nh; h++) {
        // all sign bookkeeping and pairing was done once per hypothesis in k_backward_prep: the loop below has no negations
        // (on packed operands they cost register moves), (E.x, -E.y) yields (du, dv) and (C0, -C1, -C2) directly, the accumulators
        // that receive -C1 / -C2 use negated constants (gx) or are negated once per hypothesis (G)
        const f4* sp = reinterpret_cast<const f4*>(s_rec + h * BWD_REC);
        const f4 q0 = sp[0], q1 = sp[1], r2 = sp[2], q3 = sp[3], q4 = sp[4], q5 = sp[5];
        const f2 c0 = {q0.x, q0.y}, c1 = {q0.z, q0.w}, c2 = {q1.x, q1.y}, c3 = {q1.z, q1.w};
        const f2 r0xy = {q3.x, q3.y}, nr1xy = {q3.z, q3.w}, nr2xy = {q4.x, q4.y};
        const float r0z = q4.z, nr1z = q4.w, nr2z = q5.x;
        // G pairs: Ga[i] = C_i * (X, Y)  ;  Gb[i] = C_i * (Z, 1)
        f2 Ga[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}}, Gb[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < PXG; j++) {
            float wv[PXL];
            if (!SOFTMODE) {
                if (valid[j]) {
                    if (VEC) {
                        const f4 d = __builtin_nontemporal_load(reinterpret_cast<const f4*>(d_err + (size_t)(h0 + h) * P + pbase[j]));
                        wv[0] = d.x; wv[1 % PXL] = d.y; wv[2 % PXL] = d.z; wv[3 % PXL] = d.w;
                    } else {
                        wv[0] = __builtin_nontemporal_load(d_err + (size_t)(h0 + h) * P + pbase[j]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < PXL; k++) wv[k] = 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < PXL; k++) {
                const int i = j * PXL + k;
                const float Xi = xy[i].x, Yi = xy[i].y, Zi = zw[i].x;
                // (E.x, -E.y) as one packed chain, E.z scalar
                const f2 exy = __builtin_elementwise_fma(c0, f2{Xi, Xi}, __builtin_elementwise_fma(c1, f2{Yi, Yi}, __builtin_elementwise_fma(c2, f2{Zi, Zi}, c3)));
                const float ez = fmaf(r2.x, Xi, fmaf(r2.y, Yi, fmaf(r2.z, Zi, r2.w)));
                // guard |E.z| < 1e-8 -> 0 (cnn_softam.h:416,476): a zero reciprocal keeps everything below finite
                const float iz = (fabsf(ez) >= 1e-8f) ? __builtin_amdgcn_rcpf(ez) : 0.f;
                const float fz = f * iz;
                // (u - px, v - py) with px = -f E.x/E.z + cx, py = f E.y/E.z + cy
                const f2 d = __builtin_elementwise_fma(exy, f2{fz, fz}, ppix[i]);   // exy = (E.x, -E.y)
                const f2 dq = d * d;
                const float err = __builtin_amdgcn_sqrtf(dq.x + dq.y);
                float w;
                if (SOFTMODE) {
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kA, fminf(err, clampv), kB)));
                    w = s_g[h] * (-beta) * sg * (1.0f - sg);
                } else {
                    w = wv[k];
                }
                // guards: invalid lane, E.z ~ 0 (iz == 0), err > CNN_OBJ_MAXINPUT -> zero contribution
                const bool keep = valid[j] && (iz != 0.f) && !(err > clampv);
                const float ie = __builtin_amdgcn_rcpf(err + 1e-8f);
                const float wfz = keep ? w * fz * ie : 0.f;       // w f / (E.z (err + eps))
                // a = -(du, dv)/(err+eps);  C0 = -a0 f/E.z ; C1 = a1 f/E.z ; C2 = (a0 E.x - a1 E.y) f/E.z^2   (all times w)
                const f2 Cn = d * f2{wfz, wfz};                   // (C0, -C1)
                const f2 de = d * exy;                            // (du E.x, -dv E.y)
                const float nC2 = (de.x + de.y) * (wfz * iz);     // -C2
                gxy[i] = __builtin_elementwise_fma(r0xy, f2{Cn.x, Cn.x}, __builtin_elementwise_fma(nr1xy, f2{Cn.y, Cn.y}, __builtin_elementwise_fma(nr2xy, f2{nC2, nC2}, gxy[i])));
                gz[i] = fmaf(r0z, Cn.x, fmaf(nr1z, Cn.y, fmaf(nr2z, nC2, gz[i])));
                Ga[0] = __builtin_elementwise_fma(f2{Cn.x, Cn.x}, xy[i], Ga[0]); Gb[0] = __builtin_elementwise_fma(f2{Cn.x, Cn.x}, zw[i], Gb[0]);
                Ga[1] = __builtin_elementwise_fma(f2{Cn.y, Cn.y}, xy[i], Ga[1]); Gb[1] = __builtin_elementwise_fma(f2{Cn.y, Cn.y}, zw[i], Gb[1]);   // -C1 sums
                Ga[2] = __builtin_elementwise_fma(f2{nC2, nC2}, xy[i], Ga[2]);   Gb[2] = __builtin_elementwise_fma(f2{nC2, nC2}, zw[i], Gb[2]);     // -C2 sums
            }
        }
        float G[12] = {Ga[0].x, Ga[0].y, Gb[0].x, -Ga[1].x, -Ga[1].y, -Gb[1].x, -Ga[2].x, -Ga[2].y, -Gb[2].x, Gb[0].y, -Gb[1].y, -Gb[2].y};
        const float tot = wave_sum12(G, lane);
        if (gslot >= 0) gout[(size_t)h * 12 + gslot] = tot;
    }

#pragma unroll
    for (int j = 0; j < PXG; j++) {
        if (!valid[j]) continue;
        float* dst = grad_part + (size_t)ht * P * 3 + (size_t)pbase[j] * 3;
        if (VEC) {
            f4* d4 = reinterpret_cast<f4*>(dst);
            d4[0] = f4{gxy[j * 4].x, gxy[j * 4].y, gz[j * 4], gxy[j * 4 + 1].x};
            d4[1] = f4{gxy[j * 4 + 1].y, gz[j * 4 + 1], gxy[j * 4 + 2].x, gxy[j * 4 + 2].y};
            d4[2] = f4{gz[j * 4 + 2], gxy[j * 4 + 3].x, gxy[j * 4 + 3].y, gz[j * 4 + 3]};
        } else {
            dst[0] = gxy[j].x; dst[1] = gxy[j].y; dst[2] = gz[j];
        }
    }
}


// Hypothesis tile of a launch (a kernel argument, <= K4_HT_MAX).  32 everywhere: a round-counting model (workgroups / (2 per CU),
// cost ~ rounds x HT) suggested 86 for N = 256 on a 640 x 480 map (450 workgroups in one round instead of 1200 in 2.34) and 103 for
// N = 1024; measured, 86 gains 3 % at N = 256 (175 -> 169 us) and 103 loses 5 % at N = 1024 (575 -> 602 us) -- workgroups do not
// run in lock-step rounds, and long tiles have the longer tail.
int backward_hyp_tile(int N, int P) {
    (void)N; (void)P;
    return 32;
}

int backward_num_partial_rows(int P) { return ((P + K4_THREADS - 1) / K4_THREADS) * (K4_THREADS / 64); }  // upper bound (scalar path)

hipError_t score_backward(hipStream_t st, int N, const float* staged_bwd, const FrameDev& F, const float* d_err, const double* g, float clampv,
                          float tau, float beta, float* grad_part, float* G12_part, int* partial_rows_used, int HT) {
    if (partial_rows_used) *partial_rows_used = 0;
    if (N <= 0) return hipSuccess;
    if (HT < 1 || HT > K4_HT_MAX) return hipErrorInvalidValue;
    const bool soft = d_err == nullptr;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(grad_part) & 15) == 0);
    const int tile = vec ? K4_THREADS * 4 * K4_PXG : K4_THREADS;
    const int PT = (F.P + tile - 1) / tile;
    const int NT = (N + HT - 1) / HT;
    const int grid = ((PT + 7) / 8) * 8 * NT;
    if (partial_rows_used) *partial_rows_used = PT * (K4_THREADS / 64);
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool UV = F.uv != nullptr;
#define DSAC_K4(G_, V_, S_, U_)                                                                                                              \
    hipLaunchKernelGGL((k_score_backward<G_, V_, S_, U_>), dim3(grid), dim3(K4_THREADS), 0, st, staged_bwd, F.xyz, F.uv, d_err, g, grad_part, \
                       G12_part, N, F.P, F.W, PT, NT, F.fx, F.cx, F.cy, clampv, kA, kB, beta, HT)
    if (vec) {
        if (soft) { if (UV) DSAC_K4(K4_PXG, true, true, true); else DSAC_K4(K4_PXG, true, true, false); }
        else { if (UV) DSAC_K4(K4_PXG, true, false, true); else DSAC_K4(K4_PXG, true, false, false); }
    } else {
        if (soft) { if (UV) DSAC_K4(1, false, true, true); else DSAC_K4(1, false, true, false); }
        else { if (UV) DSAC_K4(1, false, false, true); else DSAC_K4(1, false, false, false); }
    }
#undef DSAC_K4
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// finish (a): grad_xyz (double) += sum over hypothesis tiles of grad_part; quirk 1 transposes the pixel index
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grad_reduce(int P, int W, int H, int hyp_tiles, const float* __restrict__ grad_part, unsigned flags,
                                                     double* __restrict__ grad_xyz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)P * 3) return;
    double s = 0;
    for (int t = 0; t < hyp_tiles; t++) s += (double)grad_part[(size_t)t * P * 3 + i];
    size_t dst = i;
    if (flags & 1u) {
        const int p = (int)(i / 3), c = (int)(i - (size_t)p * 3);
        const int y = p / W, x = p - y * W;
        dst = ((size_t)x * W + y) * 3 + c;  // core/cnn_softam.h:628  x*cols*3 + y*3   (H == W checked by the caller)
    }
    grad_xyz[dst] += s;
}

// finish (b): one WAVE per hypothesis: G12 = sum over the partial rows (lanes stride over the rows, 48 contiguous bytes per row and
// lane, fp64 accumulation, fixed butterfly -> deterministic); G6 = [G9 . dRdH, G3]; S = G6 * dPNP; lanes 0..11 scatter the 4 x 3 sums.
// (Round 1 ran this as one THREAD per hypothesis walking all partial rows serially: 201 us for 600 rows, more than the main pass.)
__global__ __launch_bounds__(256) void k_support_scatter(int N, int W, int pixel_tiles, const float* __restrict__ G12_part,
                                                         const double* __restrict__ dRdH, const double* __restrict__ dpnp,
                                                         const int32_t* __restrict__ sets, int P, unsigned flags, double* __restrict__ grad_xyz,
                                                         double* __restrict__ G6_out) {
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (h >= N) return;
    double G[12];
#pragma unroll
    for (int i = 0; i < 12; i++) G[i] = 0;
    for (int t = lane; t < pixel_tiles; t += 64) {
        const f4* src = reinterpret_cast<const f4*>(G12_part + ((size_t)t * N + h) * 12);
        const f4 a = src[0], b = src[1], c = src[2];
        G[0] += (double)a.x; G[1] += (double)a.y; G[2] += (double)a.z; G[3] += (double)a.w;
        G[4] += (double)b.x; G[5] += (double)b.y; G[6] += (double)b.z; G[7] += (double)b.w;
        G[8] += (double)c.x; G[9] += (double)c.y; G[10] += (double)c.z; G[11] += (double)c.w;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) G[i] += __shfl_xor(G[i], o, 64);
    }
    double G6[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += G[k] * dRdH[(size_t)h * 27 + i * 9 + k];
        G6[i] = s;
        G6[3 + i] = G[9 + i];
    }
    if (G6_out && lane < 6) {
        double v = G6[0];
#pragma unroll
        for (int i = 1; i < 6; i++) v = (lane == i) ? G6[i] : v;
        G6_out[(size_t)h * 6 + lane] = v;
    }
    if (lane < 12) {
        const int i = lane / 3, c = lane - 3 * i;  // support point, channel
        int p = sets[(size_t)h * 4 + i];
        p = min(max(p, 0), P - 1);
        size_t base = (size_t)p * 3;
        if (flags & 1u) {
            const int y = p / W, x = p - y * W;
            base = ((size_t)x * W + y) * 3;  // core/cnn_softam.h:641
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += G6[k] * dpnp[(size_t)h * 72 + k * 12 + lane];
        atomicAdd(&grad_xyz[base + c], s);
    }
}

hipError_t score_backward_finish(hipStream_t st, int N, const FrameDev& F, const float* grad_part, int hyp_tiles, const float* G12_part,
                                 int pixel_tiles, const double* dRdH, const double* dpnp, const int32_t* sets, unsigned flags, double* grad_xyz,
                                 double* G6_scratch) {
    if (N <= 0) return hipSuccess;
    const size_t n3 = (size_t)F.P * 3;
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, F.P, F.W, F.H, hyp_tiles, grad_part, flags, grad_xyz);
    hipLaunchKernelGGL(k_support_scatter, dim3((N + 3) / 4), dim3(256), 0, st, N, F.W, pixel_tiles, G12_part, dRdH, dpnp, sets, F.P, flags, grad_xyz,
                       G6_scratch);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// core/train_ransac_softam.cpp:344-376: path I second term + softmax backward.  One workgroup.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_path1_softmax_bwd(int N, int P, const double* __restrict__ v6, const double* __restrict__ w,
                                                           const double* __restrict__ poses, const int32_t* __restrict__ sets,
                                                           const double* __restrict__ dpnp, double* __restrict__ grad_xyz,
                                                           double* __restrict__ g) {
    __shared__ double s_buf[256];
    const int tid = threadIdx.x;
    double v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v6[k];
    // F_h = v6 . [rvec_h ; tvec_h / 1000],  mean = sum_h w_h F_h
    double part = 0;
    for (int h = tid; h < N; h += 256) {
        double F = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) F += v[k] * poses[(size_t)h * 6 + k];
#pragma unroll
        for (int k = 3; k < 6; k++) F += v[k] * (poses[(size_t)h * 6 + k] / 1000);
        g[h] = F;  // temp
        part += w[h] * F;
    }
    s_buf[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_buf[tid] += s_buf[tid + o];
        __syncthreads();
    }
    const double mean = s_buf[0];
    for (int h = tid; h < N; h += 256) g[h] = w[h] * g[h] - w[h] * mean;
    // path I: grad[support px] += v6 . (w_h dPNP_h)
    if (dpnp && grad_xyz) {
        for (int idx = tid; idx < N * 12; idx += 256) {
            const int h = idx / 12, j = idx - h * 12;
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += v[k] * dpnp[(size_t)h * 72 + k * 12 + j];
            s *= w[h];
            int p = sets[(size_t)h * 4 + j / 3];
            p = min(max(p, 0), P - 1);
            atomicAdd(&grad_xyz[(size_t)p * 3 + (j % 3)], s);
        }
    }
}

hipError_t path1_softmax_backward(hipStream_t st, int N, int P, const double* v6, const double* w, const double* poses, const int32_t* sets,
                                  const double* dpnp, double* grad_xyz, double* g) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_path1_softmax_bwd, dim3(1), dim3(256), 0, st, N, P, v6, w, poses, sets, dpnp, grad_xyz, g);
    return hipGetLastError();
}

}  // namespace dk

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
                                                      double* __restrict__ dRdH, float f) {
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
    // LDS image of the matrix-core main pass, per group of 16 hypotheses 384 floats behind the N records: the three MFMA B operands in lane
    // order [comp][k][col] (x row, negated y row, z row) and the 16 x 12 gradient coefficients (r0, -r1, -r2 rows of R', 3 spare); the
    // last hypothesis also fills the unused columns of its group (their contributions are masked in the kernel)
    {
        float* img = rec + (size_t)N * BWD_REC;
        const int c_last = (h == N - 1) ? 15 : (h & 15);
        for (int c = h & 15; c <= c_last; c++) {
            float* gb = img + (size_t)(h >> 4) * 384;
#pragma unroll
            // the x and negated-y rows carry the focal length (round 3): the MFMA then yields f E.x and -f E.y, which is what the residual uses
            for (int k = 0; k < 4; k++) { gb[k * 16 + c] = f * o[2 * k]; gb[64 + k * 16 + c] = f * o[2 * k + 1]; gb[128 + k * 16 + c] = o[8 + k]; }
            float* gc = gb + 192 + c * 12;
            // r0 and -r1 rows carry the focal length as well: the kernel forms C0 / f and -C1 / f (one multiply per pair less, round 3)
            gc[0] = f * o[12]; gc[1] = f * o[13]; gc[2] = f * o[18]; gc[3] = f * o[14]; gc[4] = f * o[15]; gc[5] = f * o[19];
            gc[6] = o[16]; gc[7] = o[17]; gc[8] = o[20];
            gc[9] = gc[10] = gc[11] = 0.f;
        }
    }
    // rod = Rodrigues(R'), dRdH = d Rodrigues(rod) / d rod   (core/cnn_softam.h:505-509)
    double rod[3], Rre[9], J[27];
    dm::rodrigues_m2v(R, rod);
    dm::rodrigues_v2m<true>(rod, Rre, J);
#pragma unroll
    for (int k = 0; k < 27; k++) dRdH[(size_t)h * BWD_DRDH + k] = J[k];
    // Omega_i = (dR / d rod_i) R^T, skew-symmetric because R(rod) is a rotation for every rod.  The matrix-core main pass sums C (x) (E - t) instead
    // of C (x) X (X = R^T (E - t)): dLoss / d rod_i = sum_jm S[j][m] Omega_i[j][m], and only the antisymmetric part of S survives -- the kernel
    // does not accumulate its diagonal (3 of 12 accumulations per pixel pair, round 3)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int m = 0; m < 3; m++) {
                double v = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) v += J[i * 9 + 3 * j + k] * Rre[3 * m + k];
                dRdH[(size_t)h * BWD_DRDH + 27 + i * 9 + 3 * j + m] = v;
            }
}

hipError_t backward_prep(hipStream_t st, int N, const double* poses, const FrameDev& F, float* staged_bwd, double* dRdH) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_backward_prep, dim3((N + 63) / 64), dim3(64), 0, st, N, poses, staged_bwd, dRdH, F.fx);
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


// --------------------------------------------------------------------------------------------------
// K4 main pass, matrix-core form with per-lane hypothesis ownership (round 2).
//
// The rigid transform E = R'X + t' of 16 pixels under 16 hypotheses is three v_mfma_f32_16x16x4_f32 (exact fp32): the A operand holds
// (X, Y, Z, 1) of 16 consecutive pixels (rows), the B operands the x-, negated y- and z-rows of 16 hypotheses (columns).  Lane (g, c)
// of the wave then owns ONE hypothesis (column c) and FOUR CONSECUTIVE PIXELS (rows 4g .. 4g+3) in four adjacent registers, so
//   * d_err arrives as one dwordx4 per lane in exactly that layout and every per-pair operation runs packed over a PIXEL pair
//     (v_pk_fma_f32 / v_pk_mul_f32) with no register shuffles;
//   * the 12 per-hypothesis sums are plain register accumulators of the lane over its pixels -- the 35-instruction wave reduction
//     per (wave, hypothesis) of the VALU form is gone.  They are accumulated against E (already in registers) instead of X:
//     sum_p C_j X = R'^T (sum_p C_j E - t' sum_p C_j), applied once per hypothesis in the finish kernel, so the pixel coordinates are
//     not kept in registers at all.  Per 16-hypothesis group a 9-swap transpose-reduce over the 4 lane groups (v_permlane32_swap /
//     v_permlane16_swap) leaves 3 of the 12 sums in every lane, which it adds into LDS -- the workgroup is PERSISTENT over pixel
//     tiles, the LDS copy accumulates over all of them and is written out once;
//   * grad[p] += w R'^T c accumulates over the hypothesis loop in registers (12 per 16-pixel chunk) and is summed over the 16
//     hypothesis lanes of a DPP row once per pixel tile.
// A pixel tile is 64 * CH consecutive pixels (wave w: chunks w*CH .. w*CH+CH-1 of 16 pixels), a workgroup owns HT <= 256 hypotheses
// (records staged in LDS once) and walks the tiles pt0, pt0 + G, ...
//   grad_part : [hyp tile][P*3]        G12_part : [G pixel workgroups][N][12]  (E-based sums, see k_support_scatter)
// --------------------------------------------------------------------------------------------------
constexpr int K4M_HT_MAX = 256;
// K4_ABLATE (experiments only, scripts/r03_k4_ablate.sh; results are WRONG with any bit set): 1 no pose-sum accumulation, 2 no gradient
// accumulation, 4 no transpose-reduce, 8 products instead of the MFMAs, 16 no d_err stream, 32 a multiply instead of the rsq
#ifndef K4_ABLATE
#define K4_ABLATE 0
#endif

DM_INLINE float row16_sum_f(float v) {  // sum over the 16 lanes of a DPP row, result in every lane
    v += dpp_merge<0xB1, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
    v += dpp_merge<0x4E, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
    v += dpp_merge<0x141, 0xf>(0.f, v);  // row_half_mirror
    v += dpp_merge<0x140, 0xf>(0.f, v);  // row_mirror
    return v;
}

template <int CH, bool SOFTMODE, bool UV, int MINW>
__global__ __launch_bounds__(K4_THREADS, MINW) void k_score_backward_mfma(const float* __restrict__ rec, const float* __restrict__ xyz,
                                                                    const float* __restrict__ uv, const float* __restrict__ d_err,
                                                                    const double* __restrict__ g, float* __restrict__ grad_part,
                                                                    float* __restrict__ G12_part, int N, int P, int W, int PT, int NT, int G,
                                                                    float f, float cx, float cy, float clampv, float kA, float kB, float beta, int HT,
                                                                    const double* __restrict__ poses, double* __restrict__ grad_direct, unsigned gflags,
                                                                    int frame_Nf, long long xyz_stride, long long uv_stride) {
    // Round 4 (the fused stage): poses != nullptr -> the workgroup derives its hypothesis records from the cv poses itself (no k_backward_prep launch,
    // no record image in HBM) and writes G12_part hypothesis-major ([hyp][row][12]: the finish kernel reads a hypothesis' rows as one contiguous run);
    // grad_direct != nullptr (one hypothesis tile, N <= 256) -> the workgroups add their gradient straight into the caller's fp64 grad_xyz with atomics
    // (no grad_part volume, no k_grad_reduce launch; gflags bit 0 = the transposed cell index of cnn_softam.h:628).
    const int ht = blockIdx.x % NT;   // hypothesis tile of this workgroup
    const int pw = blockIdx.x / NT;   // its index among the G workgroups that share the pixel tiles
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int ngi = (nh + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, gq = lane >> 4;
    if (frame_Nf > 0) {  // frame batch: the tile's frame (no tile straddles two: HT divides the hypotheses per frame) -- its coordinate map, pixel positions and gradient
        const int fr = h0 / frame_Nf;
        xyz += (long long)fr * xyz_stride;
        if (UV) uv += (long long)fr * uv_stride;
        if (grad_direct) grad_direct += (size_t)fr * P * 3;
    }

    // LDS: MFMA B operands in lane order (x row, negated y row, z row of 16 hypotheses), the per-hypothesis coefficients of the
    // gradient accumulation, and the 12 sums of every (hypothesis, wave) accumulated over the workgroup's pixel tiles
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int HT16 = (HT + 15) >> 4;
    float* s_img = s_dyn;                                // [HT/16][384]: B operands [3][64] + coefficients [16][12] (k_backward_prep's image)
    float* s_G = s_img + HT16 * 384;                     // [HT/16][4 waves][16][12]
    {
        if (poses) {
            // one hypothesis per lane: cv -> jp pose in fp64 (dm::cv2our: Rodrigues, sign flips, det check), rounded to float and laid out as the MFMA
            // B operands [x row | negated y row | z row][k][column] and the 16 x 12 gradient coefficients -- exactly k_backward_prep's image
            for (int i = tid; i < ngi * 16; i += K4_THREADS) {
                const int hh = h0 + min(i, nh - 1);  // columns beyond the ragged end repeat the last hypothesis (masked in the loop)
                double cv6[6];
#pragma unroll
                for (int k = 0; k < 6; k++) cv6[k] = poses[(size_t)hh * 6 + k];
                double R[9], t[3];
                dm::cv2our(cv6, R, t);
                float* gb = s_img + (i >> 4) * 384;
                const int cc = i & 15;
                const float r00 = (float)R[0], r01 = (float)R[1], r02 = (float)R[2], r10 = (float)R[3], r11 = (float)R[4], r12 = (float)R[5];
                const float r20 = (float)R[6], r21 = (float)R[7], r22 = (float)R[8], t0 = (float)t[0], t1 = (float)t[1], t2 = (float)t[2];
                gb[cc] = f * r00; gb[16 + cc] = f * r01; gb[32 + cc] = f * r02; gb[48 + cc] = f * t0;
                gb[64 + cc] = f * -r10; gb[80 + cc] = f * -r11; gb[96 + cc] = f * -r12; gb[112 + cc] = f * -t1;
                gb[128 + cc] = r20; gb[144 + cc] = r21; gb[160 + cc] = r22; gb[176 + cc] = t2;
                float* gc = gb + 192 + cc * 12;
                gc[0] = f * r00; gc[1] = f * r01; gc[2] = f * r02; gc[3] = f * -r10; gc[4] = f * -r11; gc[5] = f * -r12;
                gc[6] = -r20; gc[7] = -r21; gc[8] = -r22;
                gc[9] = gc[10] = gc[11] = 0.f;
            }
        } else {
            const f4* src = reinterpret_cast<const f4*>(rec + (size_t)N * BWD_REC + (size_t)(h0 >> 4) * 384);
            f4* dst = reinterpret_cast<f4*>(s_img);
            for (int i = tid; i < ngi * 96; i += K4_THREADS) dst[i] = src[i];
        }
        f4* gz4 = reinterpret_cast<f4*>(s_G);
        for (int i = tid; i < ngi * 192; i += K4_THREADS) gz4[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (SOFTMODE) {  // dLoss/dscore of the tile's hypotheses into the spare coefficient slot
        for (int i = tid; i < nh; i += K4_THREADS) s_img[(i >> 4) * 384 + 192 + (i & 15) * 12 + 9] = (float)g[h0 + i];
        __syncthreads();
    }

    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f2 zero2 = {0.f, 0.f};
    // Work of this workgroup (round 3): a CONTIGUOUS range of (pixel tile, hypothesis group) items, the same number for every workgroup of the
    // hypothesis tile (round 2 gave whole tiles round-robin: 1200 tiles on 512 workgroups = 2 or 3 tiles each).  Measured, the balance is worth
    // nothing -- the pass is bound by the VALU issue rate of the SIMD, a workgroup that finishes early leaves its issue slots to its neighbour
    //  -- but it costs nothing either and keeps the launch time independent of how the tile count divides.  A tile whose
    // groups are split between two workgroups gets its gradient from both: the part that starts at group 0 writes layer 0 of grad_part, the part
    // that starts later writes layer 1 (a workgroup that covers a whole tile zeroes layer 1 for it), k_grad_reduce sums the layers.
    // grad_direct (round 4): the same balanced ranges; the gradient of a range's tile part is ADDED to grad_xyz with fire-and-forget fp64 atomics
    // (global_atomic_add_f64, no return value: nothing waits for them; a split tile simply has two adders).  A first version gave every workgroup
    // whole tiles and a plain read-modify-write: 1200 tiles on 512 workgroups leave 176 of them a third tile (107.7 us against 101.0), and the
    // read of the read-modify-write sits exposed at the end of every tile (profiles/r04_k4_stage.txt).
    const long long items = (long long)PT * ngi;
    long long it = items * pw / G;
    const long long it_end = items * (pw + 1) / G;
    while (it < it_end) {
        const int pt = (int)(it / ngi);
        const int g_begin = (int)(it - (long long)pt * ngi);
        const int g_end = (int)min((long long)ngi, g_begin + (it_end - it));
        it += g_end - g_begin;
        // per chunk: the MFMA A operand (coordinate gq of pixel base + c) and the position of this lane's own 4 pixels (base + 4 gq + 0..3)
        float Aop[CH];
        f2 pu[CH][2], pv[CH][2];
        int p0[CH];
        bool valid[CH];
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            const int base = (pt * (K4_THREADS / 64) * CH + wave * CH + ch) * 16;
            p0[ch] = base + 4 * gq;
            valid[ch] = p0[ch] < P;  // P % 4 == 0
            const int pa = min(base + c, P - 1);
            const float xa = xyz[(size_t)pa * 3 + min(gq, 2)];  // unconditional: a load under a lane mask makes hipcc wait vmcnt(0) at the next use of ANY load
            Aop[ch] = (gq < 3) ? xa : 1.0f;
            const int pl = min(p0[ch], P - 4);
            if (UV) {
                const f4* su = reinterpret_cast<const f4*>(uv + (size_t)pl * 2);
                const f4 u0 = su[0], u1 = su[1];
                pu[ch][0] = f2{u0.x - cx, u0.z - cx}; pv[ch][0] = f2{u0.y - cy, u0.w - cy};
                pu[ch][1] = f2{u1.x - cx, u1.z - cx}; pv[ch][1] = f2{u1.y - cy, u1.w - cy};
            } else {
                // implicit grid, W % 4 == 0 (launcher): the 4 pixels of a lane share a row
                const int y = pl / W, x = pl - y * W;
                const float xs = (float)x - cx, ys = (float)y - cy;
                pu[ch][0] = f2{xs, xs + 1.f}; pu[ch][1] = f2{xs + 2.f, xs + 3.f};
                pv[ch][0] = pv[ch][1] = f2{ys, ys};
            }
        }
        f2 gx[CH][2], gy[CH][2], gz[CH][2];
#pragma unroll
        for (int ch = 0; ch < CH; ch++)
#pragma unroll
            for (int pp = 0; pp < 2; pp++) gx[ch][pp] = gy[ch][pp] = gz[ch][pp] = zero2;

        // d_err of the first group (one dwordx4 per chunk: 4 consecutive pixels of hypothesis c)
        f4 wn[CH];
        // Every lane loads, from a clamped address (ragged hypothesis end -> the tile's last row, pixels beyond the map -> the last 4): a load under a
        // lane mask is a branch around the instruction, hipcc's wait-count pass then no longer knows how many loads are in flight and waits vmcnt(0) at
        // the first use -- which also waited for the prefetch issued just before it (every second group paid a full memory round trip; round 3,
        // profiles/r03_k4_ablate.txt: 14 us of 124).  Lanes that must not contribute are switched off through `okscale` below.
        auto load_w = [&](int gi, f4 (&dst)[CH]) {
            const int hyp = min(16 * gi + c, nh - 1);
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                if (!(K4_ABLATE & 16) && !SOFTMODE) dst[ch] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(d_err + (size_t)(h0 + hyp) * P + min(p0[ch], P - 4)));
                else dst[ch] = z4;
            }
        };
        load_w(g_begin, wn);

        // one group of 16 hypotheses; `wv` = this group's d_err (the next group's is prefetched by the loop below)
        auto group = [&](int gi, const f4 (&wv)[CH]) {
            const float* simg = s_img + gi * 384;
            const float bx = simg[lane], by = simg[64 + lane], bz = simg[128 + lane];
            const f4* cf = reinterpret_cast<const f4*>(simg + 192 + c * 12);
            const f4 c0 = cf[0], c1 = cf[1], c2 = cf[2];  // (r0.x r0.y r0.z nr1.x) (nr1.y nr1.z nr2.x nr2.y) (nr2.z g - -)
            const bool hyp_ok = 16 * gi + c < nh;
            // S[j][m] = sum_p C_j * (E.x, -E.y, E.z, 1)_m with C_0 = C0, C_1 = -C1, C_2 = -C2; the two halves = the pixel pair's lanes
            f2 S[3][4];
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) S[j][k] = zero2;
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
#if K4_ABLATE & 8
                const f4 ex4 = f4{bx, by, bz, bx} * Aop[ch], ny4 = f4{by, bz, bx, by} * Aop[ch], ez4 = f4{bz, bx, by, bz} * Aop[ch] + f4{900.f, 900.f, 900.f, 900.f};
#else
                const f4 ex4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[ch], bx, z4, 0, 0, 0);   // E.x   of pixels p0 .. p0+3, hypothesis c
                const f4 ny4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[ch], by, z4, 0, 0, 0);   // -E.y
                const f4 ez4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[ch], bz, z4, 0, 0, 0);   // E.z
#endif
                const bool lane_ok = hyp_ok && valid[ch];
                const float okscale = lane_ok ? 1e30f : 0.f;  // keep = 0 on lanes beyond the map / the ragged hypothesis end (their d_err is a clamped re-read)
#pragma unroll
                for (int pp = 0; pp < 2; pp++) {
                    const f2 ex = pp ? f2{ex4.z, ex4.w} : f2{ex4.x, ex4.y};
                    const f2 ny = pp ? f2{ny4.z, ny4.w} : f2{ny4.x, ny4.y};
                    const f2 ez = pp ? f2{ez4.z, ez4.w} : f2{ez4.x, ez4.y};
                    // One transcendental per pair instead of three (rcp E.z, sqrt, rcp err; quarter-rate instructions were a third of the
                    // pass).  With (u - px, v - py) = (A, B) / E.z,  A = (u - cx) E.z + f E.x,  B = (v - cy) E.z - f E.y,  S = A^2 + B^2:
                    //   err = sqrt(S) / |E.z|,   m = rsq(E.z^2 S) = 1 / (|E.z| sqrt(S)),   1 / E.z = m^2 S E.z,
                    //   C0 = w f A m,   -C1 = w f B m,   -C2 = w f m (A E.x - B E.y) / E.z          (px = -f E.x / E.z + cx, py = f E.y / E.z + cy)
                    const f2 A = pu[ch][pp] * ez + ex;  // ex = f E.x, ny = -f E.y (the MFMA operands carry f)
                    const f2 B = pv[ch][pp] * ez + ny;
                    const f2 Sq = __builtin_elementwise_fma(B, B, A * A);
                    const f2 zz = ez * ez;
                    // guards (cnn_softam.h:416,430,476,490): |E.z| < 1e-8 -> 0;  err > CNN_OBJ_MAXINPUT -> 0, i.e. S > clamp^2 E.z^2;  lanes beyond
                    // the map or the ragged hypothesis end -> 0;  err == 0 -> 0 (the reference divides by err + 1e-8: -0 / 1e-8)
                    // Round 3: the guards as ARITHMETIC, no compare -> scalar mask -> select chain (same instruction count, no scalar round trips;
                    // it also lets lanes beyond the map / the ragged end be switched off through the scale factor, so that their loads are unconditional):
                    //   keep = clamp((clamp^2 E.z^2 - S) * 1e30, 0, 1) is exactly 1 where err < clamp (the difference of two numbers ~1e10 is either <= 0 or
                    //   >= one ulp ~ 1e3), exactly 0 where err >= clamp -- the reference drops err > clamp only (cnn_softam.h:425,485): a cell whose fp32
                    //   residual equals CNN_OBJ_MAXINPUT to the last bit is the one measure-zero difference -- and where |E.z| < 1e-8, because then
                    //   clamp^2 E.z^2 <= 1e-12 << S = f^2 (E.x^2 + E.y^2) unless the point sits within 2e-9 mm of the camera centre in x and y as well
                    //   (no fp32 coordinate map can place one there: one ulp at 1 mm is 1e-7 mm; the reference returns 0 for it, this kernel a finite number);
                    //   m = rsq(T + 1e-30) stays finite at err == 0 (A = B = 0 there: every C is an exact 0, as the reference's 0 / 1e-8);
                    //   lanes beyond the map or the ragged hypothesis end: keep = 0 through okscale (their d_err is a clamped re-read of valid cells).
                    const f2 tk = __builtin_elementwise_fma(zz, f2{clampv * clampv, clampv * clampv}, -Sq);
                    const f2 keep = {__builtin_amdgcn_fmed3f(tk.x * okscale, 0.f, 1.f), __builtin_amdgcn_fmed3f(tk.y * okscale, 0.f, 1.f)};
                    const f2 Tp = __builtin_elementwise_fma(zz, Sq, f2{1e-30f, 1e-30f});  // E.z^2 S, kept away from 0
#if K4_ABLATE & 32
                    const f2 m = Tp * f2{1e-12f, 1e-12f};
#else
                    const f2 m = {__builtin_amdgcn_rsqf(Tp.x), __builtin_amdgcn_rsqf(Tp.y)};
#endif
                    f2 w;
                    if (SOFTMODE) {
                        const f2 err = Sq * m;  // guarded pairs: 0 -> a finite sigmoid, times m = 0 below
                        const f2 ec = {fminf(err.x, clampv), fminf(err.y, clampv)};
                        const f2 t = __builtin_elementwise_fma(f2{kA, kA}, ec, f2{kB, kB});
                        const f2 dd = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f2{1.f, 1.f};
                        const f2 sg = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
                        const float gb = c2.y * (-beta);
                        w = (sg * f2{gb, gb}) * (f2{1.f, 1.f} - sg);
                    } else {
                        w = pp ? f2{wv[ch].z, wv[ch].w} : f2{wv[ch].x, wv[ch].y};
                        if (K4_ABLATE & 16) w = f2{c2.y, c2.z} + ex;
                    }
                    w = w * keep;
                    const f2 q0 = w * m;                                  // w / (|E.z| sqrt(S))
                    const f2 iz = (m * m) * (Sq * ez);                    // 1 / E.z
                    const f2 C0 = A * q0;                                 //  C0 / f   (the factor f sits in the r0 / -r1 coefficient rows and in the finish kernel)
                    const f2 nC1 = B * q0;                                // -C1 / f
                    const f2 nC2 = __builtin_elementwise_fma(nC1, ny, C0 * ex) * iz;      // -C2 = (C0 E.x + C1 E.y) / E.z: C is orthogonal to the ray (ex, ny carry the f that C0, nC1 lack)
#if K4_ABLATE & 2
                    gx[ch][pp] += C0; gy[ch][pp] += nC1; gz[ch][pp] += nC2;
#else
                    gx[ch][pp] = __builtin_elementwise_fma(f2{c0.x, c0.x}, C0, __builtin_elementwise_fma(f2{c0.w, c0.w}, nC1, __builtin_elementwise_fma(f2{c1.z, c1.z}, nC2, gx[ch][pp])));
                    gy[ch][pp] = __builtin_elementwise_fma(f2{c0.y, c0.y}, C0, __builtin_elementwise_fma(f2{c1.x, c1.x}, nC1, __builtin_elementwise_fma(f2{c1.w, c1.w}, nC2, gy[ch][pp])));
                    gz[ch][pp] = __builtin_elementwise_fma(f2{c0.z, c0.z}, C0, __builtin_elementwise_fma(f2{c1.y, c1.y}, nC1, __builtin_elementwise_fma(f2{c2.x, c2.x}, nC2, gz[ch][pp])));
#endif
                    // E is exactly 0 where a guard fired with a NaN / inf intermediate?  No: C_j are exactly 0 there and E is finite
                    // (an MFMA of finite inputs), so the products below are exact zeros.
#if K4_ABLATE & 1
                    S[0][3] += C0; S[1][3] += nC1; S[2][3] += nC2;
#else
                    // off-diagonal sums only (the finish kernel contracts with skew-symmetric matrices, see k_backward_prep)
                    S[0][1] = __builtin_elementwise_fma(C0, ny, S[0][1]);
                    S[0][2] = __builtin_elementwise_fma(C0, ez, S[0][2]); S[0][3] += C0;
                    S[1][0] = __builtin_elementwise_fma(nC1, ex, S[1][0]);
                    S[1][2] = __builtin_elementwise_fma(nC1, ez, S[1][2]); S[1][3] += nC1;
                    S[2][0] = __builtin_elementwise_fma(nC2, ex, S[2][0]); S[2][1] = __builtin_elementwise_fma(nC2, ny, S[2][1]);
                    S[2][3] += nC2;
#endif
                }
#ifndef K4_INTERLEAVE
                __builtin_amdgcn_sched_barrier(0);  // keep the chunks apart: interleaved, their temporaries cost a wave of occupancy
#endif
            }
            // the 12 sums of hypothesis c over this lane's pixels (pair halves added), in the order a[3 j + m], a[9 + j] -> index
            // q = 0..11; then the transpose-reduce over the 4 lane groups: lane (g, c) ends with the totals of q = 3 g .. 3 g + 2
            float a[12];
#pragma unroll
            for (int j = 0; j < 3; j++) {
#pragma unroll
                for (int k = 0; k < 3; k++) a[3 * j + k] = S[j][k].x + S[j][k].y;
                a[9 + j] = S[j][3].x + S[j][3].y;
            }
#if K4_ABLATE & 4
            a[0] += a[3] + a[6] + a[9]; a[1] += a[4] + a[7] + a[10]; a[2] += a[5] + a[8] + a[11];
#else
#pragma unroll
            for (int k = 0; k < 6; k++) { lane_swap32(a[k], a[6 + k]); a[k] += a[6 + k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { lane_swap16(a[k], a[3 + k]); a[k] += a[3 + k]; }
#endif
            float* dst = s_G + ((size_t)(gi * 4 + wave) * 16 + c) * 12 + 3 * gq;  // private to this lane: plain read-modify-write
            dst[0] += a[0]; dst[1] += a[1]; dst[2] += a[2];
        };
        // Software pipeline, distance one group: this group's d_err moves out of the landing registers (4 v_mov_b64 per chunk... the wait for the loads
        // issued a whole group ago sits here), the next group's loads are issued, then the arithmetic.  The prefetch is unconditional (the last group
        // of the range re-reads its own rows: 1 / 16 of the stream, from L2) and pinned: round 3 tried two register sets that swap roles through a
        // pair of calls -- the second call sits behind a branch, LLVM then sinks the first call's loads in front of their use and every other group
        // paid the memory round trip.
        for (int gi = g_begin; gi < g_end; gi++) {
            f4 wv[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ch++) wv[ch] = wn[ch];
            load_w(min(gi + 1, g_end - 1), wn);
            __builtin_amdgcn_sched_barrier(0);
            group(gi, wv);
        }

        // grad: sum over the 16 hypothesis lanes of the row, lane c == 0 of every row stores its 4 pixels (12 consecutive floats)
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            float o[12];
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                o[6 * pp + 0] = row16_sum_f(gx[ch][pp].x); o[6 * pp + 1] = row16_sum_f(gy[ch][pp].x); o[6 * pp + 2] = row16_sum_f(gz[ch][pp].x);
                o[6 * pp + 3] = row16_sum_f(gx[ch][pp].y); o[6 * pp + 4] = row16_sum_f(gy[ch][pp].y); o[6 * pp + 5] = row16_sum_f(gz[ch][pp].y);
            }
            if (grad_direct) {
                // lane c < 12 of a row adds element c of the row's 4 pixels x 3 channels (96 contiguous bytes of fp64 per row)
                float v = o[0];
#pragma unroll
                for (int k = 1; k < 12; k++) v = (c == k) ? o[k] : v;
                if (c < 12 && valid[ch]) {
                    const int dp = (c * 11) >> 5, comp = c - 3 * dp;  // c / 3, c % 3
                    const int pix = p0[ch] + dp;
                    size_t dst = (size_t)pix * 3 + comp;
                    if (gflags & 1u) {
                        const int y = pix / W, x = pix - y * W;
                        dst = ((size_t)x * W + y) * 3 + comp;  // core/cnn_softam.h:628  x*cols*3 + y*3   (H == W checked by the caller)
                    }
                    unsafeAtomicAdd(&grad_direct[dst], (double)v);  // global_atomic_add_f64 without return
                }
            } else if (c == 0 && valid[ch]) {
                const int layer = g_begin > 0 ? 1 : 0;
                f4* dstg = reinterpret_cast<f4*>(grad_part + ((size_t)ht * 2 + layer) * P * 3 + (size_t)p0[ch] * 3);
                dstg[0] = f4{o[0], o[1], o[2], o[3]};
                dstg[1] = f4{o[4], o[5], o[6], o[7]};
                dstg[2] = f4{o[8], o[9], o[10], o[11]};
                if (g_begin == 0 && g_end == ngi) {  // the whole tile is this workgroup's: nobody writes its layer 1
                    f4* dz = reinterpret_cast<f4*>(grad_part + ((size_t)ht * 2 + 1) * P * 3 + (size_t)p0[ch] * 3);
                    dz[0] = z4; dz[1] = z4; dz[2] = z4;
                }
            }
        }
    }

    // G12 (E-based): sum over the 4 waves, one row per pixel workgroup.  [hyp][12] of a group is 192 consecutive floats in s_G as well:
    // element i of the tile lives at i + 576 * (i / 192) + 192 * wave
    __syncthreads();
    {
        const f4* sg4 = reinterpret_cast<const f4*>(s_G);
        f4* out4 = reinterpret_cast<f4*>(G12_part + ((size_t)pw * N + h0) * 12);
        f4* out4h = reinterpret_cast<f4*>(G12_part);  // hypothesis-major: [hyp][G rows][12]
        for (int i = tid; i < nh * 3; i += K4_THREADS) {
            const int j = i + 144 * (i / 48);
            const f4 v = (sg4[j] + sg4[j + 48]) + (sg4[j + 96] + sg4[j + 144]);
            if (poses) {
                const int hyp = (i * 683) >> 11, part = i - 3 * hyp;  // i / 3 for i < 768
                out4h[((size_t)(h0 + hyp) * G + pw) * 3 + part] = v;
            } else out4[i] = v;
        }
    }
}

// chunks of 16 pixels per wave and trip of the matrix-core form: variant 1..5 -> 2, 4, 5, 6, 3; the high-occupancy forms 6 -> 2 chunks at >= 4
// waves per SIMD (<= 128 VGPRs), 7 -> 3 chunks at >= 3 waves per SIMD (<= 168 VGPRs), both with 128-hypothesis tiles (37 KB of LDS: 4 workgroups per CU)
static int k4m_chunks(int variant) {
    switch (variant) {
        case 1: case 6: return 2;
        case 3: return 5;
        case 4: return 6;
        case 5: case 7: return 3;
        default: return 4;
    }
}
static int k4m_min_waves(int variant) { return variant == 6 ? 4 : variant == 7 ? 3 : 2; }

static size_t k4m_lds_bytes(int HT) {
    const int ngi = (HT + 15) / 16;
    return ((size_t)ngi * 384 + (size_t)ngi * 4 * 16 * 12) * sizeof(float);
}

// Launch plan (see kernels.h).  VALU form: hypothesis tile 32 -- a round-counting model (workgroups / (2 per CU), cost ~ rounds x HT)
// suggested 86 for N = 256 on a 640 x 480 map and 103 for N = 1024; measured, 86 gains 3 % at N = 256 and 103 loses 5 % at N = 1024
// (workgroups do not run in lock-step rounds, and long tiles have the longer tail).  Matrix-core form: all hypotheses of the frame in one
// tile up to 256 (the d_err stream is read once either way; a single tile writes grad_part once).
bool backward_variant_known(int v) {
    if (v == -1 || v == 1999) return true;
    if (v < 0) return false;
    if (v >= 1000) v -= 1000;  // + 1000: the round-3 staging (k_backward_prep + grad_part + k_grad_reduce) for A/B runs
    const int form = v % 10, tile = (v / 10) % 10, wgs = v / 100;
    return form <= 7 && tile <= 3 && wgs <= 8;
}

K4Plan backward_plan(int N, const FrameDev& F, const float* d_err, int variant, int Nf) {
    K4Plan pl{};
    pl.glayers = 1;
    const bool batch = Nf > 0 && F.frames > 1;
    // a frame's hypotheses as ONE tile up to 256, beyond that (round 6) as several equal tiles of the same launch: the largest multiple of 16 up to 256 that
    // divides the count (384 -> 192, 512 -> 256, 1024 -> 256); the tiles of a frame add into its gradient with fp64 atomics like the tiles of one big frame
    int batch_ht = 0;
    if (batch && Nf % 16 == 0) {
        for (int t = min(Nf, K4M_HT_MAX); t >= 64 || t == Nf; t -= 16)
            if (t > 0 && Nf % t == 0) { batch_ht = t; break; }
    }
    if (batch && (batch_ht == 0 || N % Nf != 0 || (variant >= 0 && variant % 10 == 0) || variant >= 1000)) { pl.variant = 0; pl.Nf = -1; return pl; }
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(F.uv) & 15) == 0) && F.P >= 4;
    // experiment knobs folded into the value: variant = form + 10 * tile code (0 auto, 1: 64, 2: 128, 3: 256) + 100 * workgroups per CU (0 auto = 2)
    int ht_code = 0, wg_per_cu = 0;
    bool legacy = false;
    if (variant >= 1000) { legacy = true; variant -= 1000; if (variant == 999) variant = -1; }  // 1999 = auto form, legacy staging
    if (variant >= 10) { wg_per_cu = variant / 100; ht_code = (variant / 10) % 10; variant = variant % 10; }
    // Chunks per wave (measured, 640 x 480, profiles/r02_k4_chunks.txt): with the d_err stream 4 (N = 256: 131 us; 5: 134, 3: 140, 6 spills),
    // without it 5 (N = 256: 144 vs 147 us, N = 1024: 485 vs 498).  A round-counting model (1200 tiles on 512 persistent workgroups =
    // 3 rounds of 4 chunks where the mean is 2.34, against 960 tiles = 2 rounds of 5) predicted -20 % for 5 chunks; it is not there:
    // a workgroup that runs out of tiles leaves its SIMDs to its neighbours.
    const bool auto_form = variant < 0;
    // round 3: 4 chunks with the d_err stream (its landing registers put the 5-chunk form over the register file: 116-152 B of scratch, 164 vs 102 us);
    // without the stream and with the implicit pixel grid 5 chunks fit exactly (256 VGPRs, no scratch): N = 256: 108 vs 115 us, N = 1024: 388 vs 400
    if (variant < 0) variant = (d_err == nullptr && F.uv == nullptr) ? 3 : 2;
    // the matrix-core form reads 4 consecutive pixels per lane as one row of the implicit grid
    if (!vec || variant > 7 || (!F.uv && F.W % 4 != 0)) variant = 0;
    pl.variant = variant;
    if (batch && variant == 0) { pl.Nf = -1; return pl; }  // the VALU fallback form has no batch mode
    if (variant == 0) {
        pl.HT = 32;
        const int tile = vec ? K4_THREADS * 4 * K4_PXG : K4_THREADS;
        pl.rows = ((F.P + tile - 1) / tile) * (K4_THREADS / 64);
    } else {
        // Small maps (the reference's 40 x 40 sub-sample: 1600 cells): with the big-map tiles the whole pass is 5 workgroups that each walk
        // 16 hypothesis groups -- 37 us of pure latency for 0.4 M pairs.  When the tiles would not even give every CU a workgroup, take 2
        // chunks per wave and the largest hypothesis tile that does (down to one 16-hypothesis group per workgroup).
        int ht_small = 0;
        if (auto_form && ht_code == 0 && !batch) {
            const int CHb = k4m_chunks(variant), PTb = (F.P + 64 * CHb - 1) / (64 * CHb);
            const int HTb = min(K4M_HT_MAX, ((max(N, 1) + 15) / 16) * 16), NTb = (max(N, 1) + HTb - 1) / HTb;
            if (PTb * NTb < 256) {
                variant = 1;
                pl.variant = 1;
                const int PT2 = (F.P + 127) / 128;
                for (ht_small = K4M_HT_MAX; ht_small > 16; ht_small >>= 1)
                    if (PT2 * ((max(N, 1) + ht_small - 1) / ht_small) >= 256) break;
            }
        }
        const int CH = k4m_chunks(variant);
        // Persistent workgroups, 2 per CU (register- and LDS-limited: 2 waves per SIMD), shared between the hypothesis tiles.  The biggest
        // hypothesis tile wins: a round-counting model preferred HT = 128 for N = 256 on 640 x 480 (5 rounds of 128 instead of 3 of 256
        // for the 1200 pixel tiles on 512 workgroups), measured it loses (145 vs 133 us; N = 1024: 541 vs 506) -- a workgroup that runs out
        // of tiles early leaves the VALU to its SIMD neighbours, so the imbalance costs far less than the extra set-up.
        const int PT = (F.P + 64 * CH - 1) / (64 * CH);
        const bool hi_occ = variant >= 6;
        const int ht_max = ht_code == 1 ? 64 : (ht_code == 2 || (hi_occ && ht_code == 0)) ? 128 : K4M_HT_MAX;
        pl.HT = min(ht_small > 0 ? ht_small : ht_max, ((max(N, 1) + 15) / 16) * 16);
        if (batch) pl.HT = batch_ht;  // one hypothesis tile per frame up to 256 hypotheses, several equal ones beyond
        pl.NT = (max(N, 1) + pl.HT - 1) / pl.HT;
        pl.rows = max(1, min(PT, ((wg_per_cu > 0 ? wg_per_cu : hi_occ ? k4m_min_waves(variant) : 2) * 256 + pl.NT - 1) / pl.NT));
        // every workgroup takes the same number of (tile, 16-hypothesis group) items: at least one group each
        pl.rows = (int)max(1ll, min((long long)pl.rows, (long long)PT * ((pl.HT + 15) / 16)));
        pl.glayers = 2;
        // round 4: the workgroups derive their records from the poses (no prep launch), G12_part is hypothesis-major, and with a single hypothesis
        // tile the gradient goes straight into grad_xyz (whole pixel tiles per workgroup: no more workgroups than tiles)
        pl.fused = !legacy;
        pl.direct = pl.fused && (pl.NT == 1 || batch);
        pl.Nf = batch ? Nf : 0;
        return pl;
    }
    pl.NT = (max(N, 1) + pl.HT - 1) / pl.HT;
    return pl;
}

hipError_t score_backward(hipStream_t st, int N, const float* staged_bwd, const FrameDev& F, const float* d_err, const double* g, float clampv,
                          float tau, float beta, float* grad_part, float* G12_part, const K4Plan& plan, const double* poses, double* grad_xyz,
                          unsigned flags) {
    if (N <= 0) return hipSuccess;
    const int HT = plan.HT, NT = plan.NT;
    const bool soft = d_err == nullptr;
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool UV = F.uv != nullptr;
    if (plan.variant > 0) {
        if (HT < 16 || HT > K4M_HT_MAX || (reinterpret_cast<uintptr_t>(grad_part) & 15)) return hipErrorInvalidValue;
        if (plan.fused && !poses) return hipErrorInvalidValue;
        if (plan.direct && !grad_xyz) return hipErrorInvalidValue;
        const double* k_poses = plan.fused ? poses : nullptr;
        double* k_direct = plan.direct ? grad_xyz : nullptr;
        const int CH = k4m_chunks(plan.variant);
        const int PT = (F.P + 64 * CH - 1) / (64 * CH);
        const int G = plan.rows;
        const int grid = G * NT;
        const size_t lds = k4m_lds_bytes(HT);
#define DSAC_K4M(C_, S_, U_, W_)                                                                                                           \
    do {                                                                                                                                    \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_backward_mfma<C_, S_, U_, W_>),                            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                            \
        if (e_ != hipSuccess) return e_;                                                                                                    \
        hipLaunchKernelGGL((k_score_backward_mfma<C_, S_, U_, W_>), dim3(grid), dim3(K4_THREADS), lds, st, staged_bwd, F.xyz, F.uv, d_err, g,   \
                           grad_part, G12_part, N, F.P, F.W, PT, NT, G, F.fx, F.cx, F.cy, clampv, kA, kB, beta, HT, k_poses, k_direct, flags,   \
                           plan.Nf, F.xyz_stride, F.uv_stride);                                                                                  \
    } while (0)
#define DSAC_K4M_CH(C_, W_)                                                                              \
    do {                                                                                                 \
        if (soft) { if (UV) DSAC_K4M(C_, true, true, W_); else DSAC_K4M(C_, true, false, W_); }          \
        else { if (UV) DSAC_K4M(C_, false, true, W_); else DSAC_K4M(C_, false, false, W_); }             \
    } while (0)
        if (plan.variant == 6) DSAC_K4M_CH(2, 4);
        else if (plan.variant == 7) DSAC_K4M_CH(3, 3);
        else switch (CH) {
            case 2: DSAC_K4M_CH(2, 2); break;
            case 3: DSAC_K4M_CH(3, 2); break;
            case 4: DSAC_K4M_CH(4, 2); break;
            case 5: DSAC_K4M_CH(5, 2); break;
            default: DSAC_K4M_CH(6, 2); break;
        }
#undef DSAC_K4M_CH
#undef DSAC_K4M
        return hipGetLastError();
    }
    if (HT < 1 || HT > K4_HT_MAX) return hipErrorInvalidValue;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(grad_part) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.uv) & 15) == 0) && F.P >= 4;
    const int tile = vec ? K4_THREADS * 4 * K4_PXG : K4_THREADS;
    const int PT = (F.P + tile - 1) / tile;
    const int grid = ((PT + 7) / 8) * 8 * NT;
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
                                                         double* __restrict__ G6_out, const float* __restrict__ rec_e, float f_e,
                                                         const double* __restrict__ poses, int Nf) {
    // poses != nullptr (round 4, the fused stage): G12_part is hypothesis-major ([hyp][row][12]: a wave reads 48 contiguous bytes per lane, the
    // whole hypothesis one contiguous run), the E-based sums apply, and dR/drod, Omega and t' are derived here from the cv pose -- every lane of the
    // hypothesis' wave redundantly, under the partial-row loads -- instead of being read from k_backward_prep's output
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (h >= N) return;
    if (Nf > 0) grad_xyz += (size_t)(h / Nf) * P * 3;  // frame batch: the support cells of hypothesis h lie in frame h / Nf
    // the first 8 x 64 partial rows (all 512 of the persistent main pass) are requested at once, before the pose-only fp64 work below, which then runs
    // under their latency; rows beyond them (the VALU form's per-tile rows) follow in a plain loop
    constexpr int PRE = 8;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    f4 pre[PRE][3];
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        const int t = lane + 64 * k;
        if (t < pixel_tiles) {
            const f4* src = reinterpret_cast<const f4*>(G12_part + (poses ? ((size_t)h * pixel_tiles + t) : ((size_t)t * N + h)) * 12);
            pre[k][0] = src[0]; pre[k][1] = src[1]; pre[k][2] = src[2];
        } else {
            pre[k][0] = z4; pre[k][1] = z4; pre[k][2] = z4;
        }
    }
    double Dloc[27];
    float tpf[3] = {0.f, 0.f, 0.f};
    if (poses) {
        double cv6[6], R[9], t[3], rod[3], Rre[9], J[27];
#pragma unroll
        for (int k = 0; k < 6; k++) cv6[k] = poses[(size_t)h * 6 + k];
        dm::cv2our(cv6, R, t);
        tpf[0] = (float)t[0]; tpf[1] = (float)t[1]; tpf[2] = (float)t[2];  // t' as the fp32 numbers the main pass computed E from
        dm::rodrigues_m2v(R, rod);
        dm::rodrigues_v2m<true>(rod, Rre, J);  // rod = Rodrigues(R'), J = d Rodrigues(rod) / d rod   (core/cnn_softam.h:505-509)
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int m = 0; m < 3; m++) {  // Omega_i = (dR / d rod_i) R^T (see k_backward_prep)
                    double v = 0;
#pragma unroll
                    for (int k = 0; k < 3; k++) v += J[i * 9 + 3 * j + k] * Rre[3 * m + k];
                    Dloc[i * 9 + 3 * j + m] = v;
                }
    }
    double G[12];
#pragma unroll
    for (int i = 0; i < 12; i++) G[i] = 0;
#pragma unroll
    for (int k = 0; k < PRE; k++) {  // same order of additions as one loop over t = lane, lane + 64, ...
        const f4 a = pre[k][0], b = pre[k][1], c = pre[k][2];
        G[0] += (double)a.x; G[1] += (double)a.y; G[2] += (double)a.z; G[3] += (double)a.w;
        G[4] += (double)b.x; G[5] += (double)b.y; G[6] += (double)b.z; G[7] += (double)b.w;
        G[8] += (double)c.x; G[9] += (double)c.y; G[10] += (double)c.z; G[11] += (double)c.w;
    }
    for (int t = lane + 64 * PRE; t < pixel_tiles; t += 64) {
        const f4* src = reinterpret_cast<const f4*>(G12_part + (poses ? ((size_t)h * pixel_tiles + t) : ((size_t)t * N + h)) * 12);
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
    if (rec_e || poses) {
        // matrix-core main pass: the sums were taken against E = R'X + t' (signed as the kernel held them):
        //   a[3j + m] = sum C~_j E~_m, a[9 + j] = sum C~_j  with C~ = (C0, -C1, -C2), E~ = (E.x, -E.y, E.z)
        // sum_p C_j (E - t')_m = sum C_j E_m - t'_m sum C_j  (t' as the fp32 record the kernel computed E from); X = R^T (E - t) is applied through
        // Omega below
        double tp[3];
        if (poses) { tp[0] = tpf[0]; tp[1] = tpf[1]; tp[2] = tpf[2]; }
        else {
            const float* o = rec_e + (size_t)h * BWD_REC;
            tp[0] = o[6]; tp[1] = -(double)o[7]; tp[2] = o[11];
        }
        // the kernel summed (C0 / f, -C1 / f, -C2) against (f E.x, -f E.y, E.z), off-diagonal pairs only
        const double sj[3] = {(double)f_e, -(double)f_e, -1}, sm[3] = {1.0 / (double)f_e, -1.0 / (double)f_e, 1};
        double A[3][3], B[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            B[j] = sj[j] * G[9 + j];
#pragma unroll
            for (int m = 0; m < 3; m++) A[j][m] = sj[j] * sm[m] * G[3 * j + m];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int m = 0; m < 3; m++) G[3 * j + m] = (j == m) ? 0.0 : A[j][m] - tp[m] * B[j];   // sum_p C_j (E - t)_m, j != m
            G[9 + j] = B[j];
        }
    }
    // rotation part: sum C_j X_k against dR/drod (VALU form), or sum C_j (E - t)_m against Omega = dR/drod R^T (matrix-core form)
    const double* D = poses ? nullptr : dRdH + (size_t)h * BWD_DRDH + (rec_e ? 27 : 0);
    double G6[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += G[k] * (poses ? Dloc[i * 9 + k] : D[i * 9 + k]);
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
                                 double* G6_scratch, const float* rec_if_e_based, const double* poses_if_fused, int Nf) {
    if (N <= 0) return hipSuccess;
    const size_t n3 = (size_t)F.P * 3;
    // hyp_tiles == 0: the main pass has added its gradient into grad_xyz itself (K4Plan.direct)
    if (hyp_tiles > 0) hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, F.P, F.W, F.H, hyp_tiles, grad_part, flags, grad_xyz);
    hipLaunchKernelGGL(k_support_scatter, dim3((N + 3) / 4), dim3(256), 0, st, N, F.W, pixel_tiles, G12_part, dRdH, dpnp, sets, F.P, flags, grad_xyz,
                       G6_scratch, rec_if_e_based, F.fx, poses_if_fused, F.frames > 1 ? Nf : 0);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K4 PARITY MODE (fp64, the reference's own evaluation order): dScore part (iii) exactly as core/cnn_softam.h:609-645 runs it -- one hypothesis
// after the other, cells in the reference's loop order (x outer, y inner), dProjectdObj (:404-453) and dProjectdHyp (:464-528) in double, the
// per-hypothesis Jacobian kept (copyTo semantics: a cell's direct term overwrites, the support term is added after the loop) and summed over the
// hypotheses in index order.  With DSAC_BWD_QUIRK_ROT_WRITEBACK the rotation matrix is re-derived from its own Rodrigues vector and WRITTEN BACK on
// every dProjectdHyp call (quirk 7: the reference does that through a const reference, :506-508), so it drifts by round-off from cell to cell --
// a sequential recurrence, which is why this mode runs one LANE per hypothesis.  It exists for parity (tests against the real reference's golden
// vectors to 1e-9), not for speed: reference-sized maps only (N * H * W bounded by the caller).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_score_backward_parity(int N, int H, int W, const double* __restrict__ poses, const float* __restrict__ xyz,
                                                              const float* __restrict__ uv, const float* __restrict__ d_err,
                                                              const double* __restrict__ dpnp, const int32_t* __restrict__ sets, double f, double cx,
                                                              double cy, unsigned flags, double* __restrict__ jac /* N x P*3 */,
                                                              double* __restrict__ G6_out) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    const int P = H * W;
    const bool transpose = flags & 1u, writeback = flags & 4u;
    double cv6[6], R[9], t[3];
#pragma unroll
    for (int k = 0; k < 6; k++) cv6[k] = poses[(size_t)h * 6 + k];
    dm::cv2our(cv6, R, t);
    double* J = jac + (size_t)h * P * 3;
    for (int i = 0; i < P * 3; i++) J[i] = 0.0;
    double G6[6] = {0, 0, 0, 0, 0, 0};
    const double EPS = 1e-8, MAXE = 100.0;  // core/cnn_softam.h:43 EPS, core/lua_calls.h:36 CNN_OBJ_MAXINPUT
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            const int p = y * W + x;
            const double X0 = xyz[(size_t)p * 3], X1 = xyz[(size_t)p * 3 + 1], X2 = xyz[(size_t)p * 3 + 2];
            const double u = uv ? (double)uv[(size_t)p * 2] : (double)x, v = uv ? (double)uv[(size_t)p * 2 + 1] : (double)y;
            const double w = (double)d_err[(size_t)h * P + p];
            const int col = transpose ? (x * W * 3 + y * 3) : (p * 3);
            // ---- dProjectdObj with the CURRENT rotation (the write-back of the previous cell's dProjectdHyp is in effect)
            {
                const double E0 = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0], E1 = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1],
                             E2 = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
                double d[3] = {0, 0, 0};
                if (!(fabs(E2) < EPS)) {
                    const double px = -f * E0 / E2 + cx, py = f * E1 / E2 + cy;
                    double err = sqrt((u - px) * (u - px) + (v - py) * (v - py));
                    if (!(err > MAXE)) {
                        err += EPS;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const double pxd = -f * R[c] / E2 + f * E0 / E2 / E2 * R[6 + c];
                            const double pyd = f * R[3 + c] / E2 - f * E1 / E2 / E2 * R[6 + c];
                            d[c] = 0.5 / err * (2 * (u - px) * -pxd + 2 * (v - py) * -pyd);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; c++) J[col + c] = w * d[c];  // copyTo: overwrite
            }
            // ---- dProjectdHyp
            {
                const double E0 = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0], E1 = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1],
                             E2 = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
                if (fabs(E2) < EPS) continue;
                const double px = -f * E0 / E2 + cx, py = f * E1 / E2 + cy;
                double err = sqrt((u - px) * (u - px) + (v - py) * (v - py));
                if (err > MAXE) continue;
                err += EPS;
                const double n0 = -1 / err * (u - px), n1 = -1 / err * (v - py);
                const double Xv[3] = {X0, X1, X2};
                double q9[9];  // dNdP . dPdR  (1 x 9)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    q9[c] = n0 * (-f * Xv[c] / E2);
                    q9[3 + c] = n1 * (f * Xv[c] / E2);
                    q9[6 + c] = n0 * (f * E0 / E2 / E2 * Xv[c]) + n1 * (-f * E1 / E2 / E2 * Xv[c]);
                }
                double rod[3], Rre[9], Jr[27];
                dm::rodrigues_m2v(R, rod);
                dm::rodrigues_v2m<true>(rod, Rre, Jr);
                if (writeback) {
#pragma unroll
                    for (int k = 0; k < 9; k++) R[k] = Rre[k];
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += q9[k] * Jr[i * 9 + k];
                    G6[i] += w * s;
                }
                G6[3] += w * (n0 * (-f / E2));
                G6[4] += w * (n1 * (f / E2));
                G6[5] += w * (n0 * (f * E0 / E2 / E2) + n1 * (-f * E1 / E2 / E2));
            }
        }
    // support term: S = G6 . dPNP_h added to the 4 cells of the minimal set (:636-642)
    for (int i = 0; i < 4; i++) {
        int p = sets[(size_t)h * 4 + i];
        p = min(max(p, 0), P - 1);
        const int y = p / W, x = p - y * W;
        const int col = transpose ? (x * W * 3 + y * 3) : (p * 3);
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += G6[k] * dpnp[(size_t)h * 72 + k * 12 + i * 3 + c];
            J[col + c] += s;
        }
    }
    if (G6_out)
        for (int k = 0; k < 6; k++) G6_out[(size_t)h * 6 + k] = G6[k];
}

// grad_xyz += sum over hypotheses, in index order (core/train_ransac_softam.cpp:382-383)
__global__ __launch_bounds__(256) void k_parity_sum(int N, size_t n3, const double* __restrict__ jac, double* __restrict__ grad_xyz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    double s = 0;
    for (int h = 0; h < N; h++) s += jac[(size_t)h * n3 + i];
    grad_xyz[i] += s;
}

hipError_t score_backward_parity(hipStream_t st, int N, const double* poses, const FrameDev& F, const float* d_err, const double* dpnp, const int32_t* sets,
                                 unsigned flags, double* jac_scratch, double* grad_xyz, double* G6_out) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_score_backward_parity, dim3((N + 63) / 64), dim3(64), 0, st, N, F.H, F.W, poses, F.xyz, F.uv, d_err, dpnp, sets, (double)F.fx,
                       (double)F.cx, (double)F.cy, flags, jac_scratch, G6_out);
    const size_t n3 = (size_t)F.P * 3;
    hipLaunchKernelGGL(k_parity_sum, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, N, n3, jac_scratch, grad_xyz);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// core/train_ransac_softam.cpp:344-376: path I second term + softmax backward.  One workgroup.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_path1_softmax_bwd(int N, int P, const double* __restrict__ v6, const double* __restrict__ w,
                                                           const double* __restrict__ poses, const int32_t* __restrict__ sets,
                                                           const double* __restrict__ dpnp, double* __restrict__ grad_xyz,
                                                           double* __restrict__ g, double g_scale) {
    {   // frame f of a batch (blockIdx.x): N hypotheses per frame
        const size_t f = blockIdx.x;
        v6 += 6 * f; w += f * N; poses += f * N * 6; g += f * N;
        if (sets) sets += f * N * 4;
        if (dpnp) dpnp += f * N * 72;
        if (grad_xyz) grad_xyz += f * (size_t)P * 3;
    }
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
    for (int h = tid; h < N; h += 256) g[h] = (w[h] * g[h] - w[h] * mean) * g_scale;
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
                                  const double* dpnp, double* grad_xyz, double* g, double g_scale, int frames) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_path1_softmax_bwd, dim3(frames < 1 ? 1 : frames), dim3(256), 0, st, N, P, v6, w, poses, sets, dpnp, grad_xyz, g, g_scale);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// core/train_ransac_softam.cpp:322 and :350-353 on the device: grad[px_i] += dL . dRefineObj_i (sparse 6 x 3 blocks, scaled by the
// caller's skip already) and v6 = dL . dRefineHyp, so that the chain dLossMax -> dRefine -> dPNP / softmax backward needs no host round trip.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_path1_assemble(const double* __restrict__ dL, const double* __restrict__ J_hyp, const int32_t* __restrict__ obj_pixels,
                                                        const double* __restrict__ J_obj, const int32_t* __restrict__ n_obj, int cap, int P,
                                                        double* __restrict__ grad_xyz, double* __restrict__ v6, int px_stride) {
    {   // frame f of a batch (blockIdx.y)
        const size_t f = blockIdx.y;
        dL += 6 * f; J_hyp += 36 * f; obj_pixels += f * (size_t)px_stride; J_obj += f * (size_t)cap * 18; n_obj += f; grad_xyz += f * (size_t)P * 3; v6 += 6 * f;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(n_obj[0], cap);
    if (i < n * 3) {
        const int cell = i / 3, c = i - cell * 3;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += dL[k] * J_obj[(size_t)cell * 18 + k * 3 + c];
        const int p = min(max(obj_pixels[cell], 0), P - 1);
        grad_xyz[(size_t)p * 3 + c] += s;  // the selected cells are distinct
    }
    if (blockIdx.x == 0 && threadIdx.x < 6) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += dL[k] * J_hyp[k * 6 + threadIdx.x];
        v6[threadIdx.x] = s;
    }
}

hipError_t path1_assemble(hipStream_t st, const double* dL, const double* J_hyp, const int32_t* obj_pixels, const double* J_obj, const int32_t* n_obj,
                          int cap, int P, double* grad_xyz, double* v6, int frames, int px_stride) {
    hipLaunchKernelGGL(k_path1_assemble, dim3((cap * 3 + 255) / 256 + 1, frames < 1 ? 1 : frames), dim3(256), 0, st, dL, J_hyp, obj_pixels, J_obj, n_obj, cap, P,
                       grad_xyz, v6, px_stride > 0 ? px_stride : cap);
    return hipGetLastError();
}

}  // namespace dk

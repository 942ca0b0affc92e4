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

constexpr int K4_THREADS = 256;
constexpr int K4_HT = 32;
constexpr int BWD_REC = 12;  // floats per hypothesis: R'0|t'0, R'1|t'1, R'2|t'2

int backward_hyp_tile() { return K4_HT; }

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
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o[i * 4 + 0] = (float)R[i * 3 + 0]; o[i * 4 + 1] = (float)R[i * 3 + 1]; o[i * 4 + 2] = (float)R[i * 3 + 2];
        o[i * 4 + 3] = (float)t[i];
    }
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
// Sum over the 64 lanes with DPP only (no LDS crossbar): 4 intra-row steps + row_bcast:15 / row_bcast:31.
// The total is valid in lane 63.
DM_INLINE float wave_sum_to_lane63(float v) {
#define DSAC_DPP_ADD(ctrl, rmask)                                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    DSAC_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
    DSAC_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
    DSAC_DPP_ADD(0x141, 0xf);  // row_half_mirror
    DSAC_DPP_ADD(0x140, 0xf);  // row_mirror        -> every lane of a row holds the row sum
    DSAC_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    DSAC_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> row 3 holds the wave sum
#undef DSAC_DPP_ADD
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
                                                               float f, float cx, float cy, float clampv, float kA, float kB, float beta) {
    constexpr int PXL = VEC ? 4 : 1;          // pixels per group
    constexpr int NP = PXG * PXL;             // pixels per lane
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ht = q % NT;
    const int pt = (q / NT) * 8 + (b & 7);
    if (pt >= PT) return;
    const int h0 = ht * K4_HT;
    const int nh = min(K4_HT, N - h0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    __shared__ __attribute__((aligned(16))) float s_rec[K4_HT * BWD_REC];
    __shared__ float s_g[K4_HT];
    for (int i = tid; i < nh * BWD_REC; i += K4_THREADS) s_rec[i] = rec[(size_t)h0 * BWD_REC + i];
    if (SOFTMODE && tid < nh) s_g[tid] = (float)g[h0 + tid];

    const int tile0 = pt * K4_THREADS * NP;
    int pbase[PXG];
    bool valid[PXG];
    float X[NP], Y[NP], Z[NP], pu[NP], pv[NP];
#pragma unroll
    for (int j = 0; j < PXG; j++) {
        pbase[j] = tile0 + j * K4_THREADS * PXL + tid * PXL;
        valid[j] = pbase[j] < P;  // VEC is only used with P % 4 == 0
        if (VEC) {
            if (valid[j]) {
                const f4* src = reinterpret_cast<const f4*>(xyz + (size_t)pbase[j] * 3);
                const f4 a = src[0], bb = src[1], c = src[2];
                X[j * 4 + 0] = a.x; Y[j * 4 + 0] = a.y; Z[j * 4 + 0] = a.z; X[j * 4 + 1] = a.w; Y[j * 4 + 1] = bb.x; Z[j * 4 + 1] = bb.y;
                X[j * 4 + 2] = bb.z; Y[j * 4 + 2] = bb.w; Z[j * 4 + 2] = c.x; X[j * 4 + 3] = c.y; Y[j * 4 + 3] = c.z; Z[j * 4 + 3] = c.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) X[j * 4 + k] = Y[j * 4 + k] = Z[j * 4 + k] = 0.f;
            }
        } else {
            if (valid[j]) { X[j] = xyz[(size_t)pbase[j] * 3]; Y[j] = xyz[(size_t)pbase[j] * 3 + 1]; Z[j] = xyz[(size_t)pbase[j] * 3 + 2]; }
            else { X[j] = Y[j] = Z[j] = 0.f; }
        }
#pragma unroll
        for (int k = 0; k < PXL; k++) {
            const int p = pbase[j] + k;
            if (UV) {
                pu[j * PXL + k] = valid[j] ? uv[(size_t)p * 2] - cx : 0.f;
                pv[j * PXL + k] = valid[j] ? uv[(size_t)p * 2 + 1] - cy : 0.f;
            } else {
                const int y = p / W, x = p - y * W;
                pu[j * PXL + k] = (float)x - cx;
                pv[j * PXL + k] = (float)y - cy;
            }
        }
    }
    __syncthreads();

    float gx[NP][3];
#pragma unroll
    for (int k = 0; k < NP; k++) gx[k][0] = gx[k][1] = gx[k][2] = 0.f;

    float* gout = G12_part + ((size_t)(pt * (K4_THREADS / 64) + wave) * N + h0) * 12;
    for (int h = 0; h < nh; h++) {
        const f4* sp = reinterpret_cast<const f4*>(s_rec + h * BWD_REC);
        const f4 r0 = sp[0], r1 = sp[1], r2 = sp[2];
        float G[12];
#pragma unroll
        for (int i = 0; i < 12; i++) G[i] = 0.f;
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
                const float ex = fmaf(r0.x, X[i], fmaf(r0.y, Y[i], fmaf(r0.z, Z[i], r0.w)));
                const float ey = fmaf(r1.x, X[i], fmaf(r1.y, Y[i], fmaf(r1.z, Z[i], r1.w)));
                const float ez = fmaf(r2.x, X[i], fmaf(r2.y, Y[i], fmaf(r2.z, Z[i], r2.w)));
                const float iz = __builtin_amdgcn_rcpf(ez);
                const float fz = f * iz;
                const float du = fmaf(ex, fz, pu[i]);    // u - px,  px = -f ex/ez + cx
                const float dv = fmaf(-ey, fz, pv[i]);   // v - py,  py =  f ey/ez + cy
                const float err = __builtin_amdgcn_sqrtf(fmaf(dv, dv, du * du));
                // guards of the reference: |E.z| < 1e-8 -> 0 ; err > CNN_OBJ_MAXINPUT -> 0
                const bool keep = valid[j] && (fabsf(ez) >= 1e-8f) && !(err > clampv);
                float w;
                if (SOFTMODE) {
                    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kA, fminf(err, clampv), kB)));
                    w = s_g[h] * (-beta) * s * (1.0f - s);
                } else {
                    w = wv[k];
                }
                const float ie = __builtin_amdgcn_rcpf(err + 1e-8f);
                const float wfz = w * fz * ie;                    // w f / (E.z (err + eps))
                // a = -(du, dv)/(err+eps);  c0 = -a0 f/E.z ; c1 = a1 f/E.z ; c2 = (a0 E.x - a1 E.y) f/E.z^2   (all times w);
                // the selects also kill the inf/NaN of E.z == 0
                const float C0 = keep ? du * wfz : 0.f;
                const float C1 = keep ? -dv * wfz : 0.f;
                const float C2 = keep ? (dv * ey - du * ex) * wfz * iz : 0.f;
                gx[i][0] = fmaf(r0.x, C0, fmaf(r1.x, C1, fmaf(r2.x, C2, gx[i][0])));
                gx[i][1] = fmaf(r0.y, C0, fmaf(r1.y, C1, fmaf(r2.y, C2, gx[i][1])));
                gx[i][2] = fmaf(r0.z, C0, fmaf(r1.z, C1, fmaf(r2.z, C2, gx[i][2])));
                G[0] = fmaf(C0, X[i], G[0]); G[1] = fmaf(C0, Y[i], G[1]); G[2] = fmaf(C0, Z[i], G[2]);
                G[3] = fmaf(C1, X[i], G[3]); G[4] = fmaf(C1, Y[i], G[4]); G[5] = fmaf(C1, Z[i], G[5]);
                G[6] = fmaf(C2, X[i], G[6]); G[7] = fmaf(C2, Y[i], G[7]); G[8] = fmaf(C2, Z[i], G[8]);
                G[9] += C0; G[10] += C1; G[11] += C2;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i++) G[i] = wave_sum_to_lane63(G[i]);
        if (lane == 63) {
            f4* o = reinterpret_cast<f4*>(gout + (size_t)h * 12);
            o[0] = f4{G[0], G[1], G[2], G[3]};
            o[1] = f4{G[4], G[5], G[6], G[7]};
            o[2] = f4{G[8], G[9], G[10], G[11]};
        }
    }

#pragma unroll
    for (int j = 0; j < PXG; j++) {
        if (!valid[j]) continue;
        float* dst = grad_part + (size_t)ht * P * 3 + (size_t)pbase[j] * 3;
        if (VEC) {
            f4* d4 = reinterpret_cast<f4*>(dst);
            d4[0] = f4{gx[j * 4][0], gx[j * 4][1], gx[j * 4][2], gx[j * 4 + 1][0]};
            d4[1] = f4{gx[j * 4 + 1][1], gx[j * 4 + 1][2], gx[j * 4 + 2][0], gx[j * 4 + 2][1]};
            d4[2] = f4{gx[j * 4 + 2][2], gx[j * 4 + 3][0], gx[j * 4 + 3][1], gx[j * 4 + 3][2]};
        } else {
            dst[0] = gx[j][0]; dst[1] = gx[j][1]; dst[2] = gx[j][2];
        }
    }
}

constexpr int K4_PXG = 2;  // 8 pixels per lane on the vector path

int backward_num_partial_rows(int P) { return ((P + K4_THREADS - 1) / K4_THREADS) * (K4_THREADS / 64); }  // upper bound (scalar path)

hipError_t score_backward(hipStream_t st, int N, const float* staged_bwd, const FrameDev& F, const float* d_err, const double* g, float clampv,
                          float tau, float beta, float* grad_part, float* G12_part, int* partial_rows_used) {
    if (partial_rows_used) *partial_rows_used = 0;
    if (N <= 0) return hipSuccess;
    const bool soft = d_err == nullptr;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(grad_part) & 15) == 0);
    const int tile = vec ? K4_THREADS * 4 * K4_PXG : K4_THREADS;
    const int PT = (F.P + tile - 1) / tile;
    const int NT = (N + K4_HT - 1) / K4_HT;
    const int grid = ((PT + 7) / 8) * 8 * NT;
    if (partial_rows_used) *partial_rows_used = PT * (K4_THREADS / 64);
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool UV = F.uv != nullptr;
#define DSAC_K4(G_, V_, S_, U_)                                                                                                              \
    hipLaunchKernelGGL((k_score_backward<G_, V_, S_, U_>), dim3(grid), dim3(K4_THREADS), 0, st, staged_bwd, F.xyz, F.uv, d_err, g, grad_part, \
                       G12_part, N, F.P, F.W, PT, NT, F.fx, F.cx, F.cy, clampv, kA, kB, beta)
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

// finish (b): one thread per hypothesis: G12 = sum over pixel tiles; G6 = [G9 . dRdH, G3]; S = G6 * dPNP; scatter.
__global__ __launch_bounds__(64) void k_support_scatter(int N, int W, int pixel_tiles, const float* __restrict__ G12_part,
                                                        const double* __restrict__ dRdH, const double* __restrict__ dpnp,
                                                        const int32_t* __restrict__ sets, int P, unsigned flags, double* __restrict__ grad_xyz,
                                                        double* __restrict__ G6_out) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    double G[12];
#pragma unroll
    for (int i = 0; i < 12; i++) G[i] = 0;
    for (int t = 0; t < pixel_tiles; t++) {
        const float* src = G12_part + ((size_t)t * N + h) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) G[i] += (double)src[i];
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
    if (G6_out) {
#pragma unroll
        for (int i = 0; i < 6; i++) G6_out[(size_t)h * 6 + i] = G6[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int p = sets[(size_t)h * 4 + i];
        p = min(max(p, 0), P - 1);
        size_t base = (size_t)p * 3;
        if (flags & 1u) {
            const int y = p / W, x = p - y * W;
            base = ((size_t)x * W + y) * 3;  // core/cnn_softam.h:641
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += G6[k] * dpnp[(size_t)h * 72 + k * 12 + i * 3 + c];
            atomicAdd(&grad_xyz[base + c], s);
        }
    }
}

hipError_t score_backward_finish(hipStream_t st, int N, const FrameDev& F, const float* grad_part, int hyp_tiles, const float* G12_part,
                                 int pixel_tiles, const double* dRdH, const double* dpnp, const int32_t* sets, unsigned flags, double* grad_xyz,
                                 double* G6_scratch) {
    if (N <= 0) return hipSuccess;
    const size_t n3 = (size_t)F.P * 3;
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, F.P, F.W, F.H, hyp_tiles, grad_part, flags, grad_xyz);
    hipLaunchKernelGGL(k_support_scatter, dim3((N + 63) / 64), dim3(64), 0, st, N, F.W, pixel_tiles, G12_part, dRdH, dpnp, sets, F.P, flags, grad_xyz,
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

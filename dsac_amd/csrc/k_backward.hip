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
constexpr int K4_PX = 4;
constexpr int K4_HT = 32;
constexpr int BWD_REC = 12;  // floats per hypothesis: R'0|t'0, R'1|t'1, R'2|t'2

int backward_num_pixel_tiles(int P) { return (P + K4_THREADS - 1) / K4_THREADS; }  // upper bound (scalar path)
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
DM_INLINE float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SOFTMODE: w = g[h] * d soft / d err, soft = sigmoid(beta (tau - min(err, clamp)))
template <int PX, bool SOFTMODE, bool UV>
__global__ __launch_bounds__(K4_THREADS) void k_score_backward(const float* __restrict__ rec, const float* __restrict__ xyz,
                                                               const float* __restrict__ uv, const float* __restrict__ d_err,
                                                               const double* __restrict__ g, float* __restrict__ grad_part,
                                                               float* __restrict__ G12_part, int N, int P, int W, int PT, int NT,
                                                               float f, float cx, float cy, float clampv, float kA, float kB, float beta) {
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
    __shared__ float s_red[(K4_THREADS / 64) * 12];
    for (int i = tid; i < nh * BWD_REC; i += K4_THREADS) s_rec[i] = rec[(size_t)h0 * BWD_REC + i];
    if (SOFTMODE && tid < nh) s_g[tid] = (float)g[h0 + tid];

    const int p0 = (pt * K4_THREADS + tid) * PX;
    const bool valid = p0 < P;
    float X[PX], Y[PX], Z[PX], pu[PX], pv[PX];
    if (PX == 4) {
        if (valid) {
            const f4* src = reinterpret_cast<const f4*>(xyz + (size_t)p0 * 3);
            const f4 a = src[0], bb = src[1], c = src[2];
            X[0] = a.x; Y[0] = a.y; Z[0] = a.z; X[1] = a.w; Y[1] = bb.x; Z[1] = bb.y;
            X[2] = bb.z; Y[2] = bb.w; Z[2] = c.x; X[3] = c.y; Y[3] = c.z; Z[3] = c.w;
            if (UV) {
                const f4* su = reinterpret_cast<const f4*>(uv + (size_t)p0 * 2);
                const f4 u0 = su[0], u1 = su[1];
                pu[0] = u0.x - cx; pv[0] = u0.y - cy; pu[1] = u0.z - cx; pv[1] = u0.w - cy;
                pu[2] = u1.x - cx; pv[2] = u1.y - cy; pu[3] = u1.z - cx; pv[3] = u1.w - cy;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PX; k++) { X[k] = Y[k] = Z[k] = 0.f; pu[k] = pv[k] = 0.f; }
        }
    } else {
        if (valid) {
            X[0] = xyz[(size_t)p0 * 3]; Y[0] = xyz[(size_t)p0 * 3 + 1]; Z[0] = xyz[(size_t)p0 * 3 + 2];
            if (UV) { pu[0] = uv[(size_t)p0 * 2] - cx; pv[0] = uv[(size_t)p0 * 2 + 1] - cy; }
        } else { X[0] = Y[0] = Z[0] = 0.f; pu[0] = pv[0] = 0.f; }
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
    __syncthreads();

    float gx[PX][3];
#pragma unroll
    for (int k = 0; k < PX; k++) gx[k][0] = gx[k][1] = gx[k][2] = 0.f;

    for (int h = 0; h < nh; h++) {
        const f4* sp = reinterpret_cast<const f4*>(s_rec + h * BWD_REC);
        const f4 r0 = sp[0], r1 = sp[1], r2 = sp[2];
        float wv[PX];
        if (!SOFTMODE) {
            if (valid) {
                if (PX == 4) {
                    const f4 d = __builtin_nontemporal_load(reinterpret_cast<const f4*>(d_err + (size_t)(h0 + h) * P + p0));
                    wv[0] = d.x; wv[1] = d.y; wv[2] = d.z; wv[3] = d.w;
                } else {
                    wv[0] = __builtin_nontemporal_load(d_err + (size_t)(h0 + h) * P + p0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < PX; k++) wv[k] = 0.f;
            }
        }
        float G[12];
#pragma unroll
        for (int i = 0; i < 12; i++) G[i] = 0.f;
#pragma unroll
        for (int k = 0; k < PX; k++) {
            const float ex = fmaf(r0.x, X[k], fmaf(r0.y, Y[k], fmaf(r0.z, Z[k], r0.w)));
            const float ey = fmaf(r1.x, X[k], fmaf(r1.y, Y[k], fmaf(r1.z, Z[k], r1.w)));
            const float ez = fmaf(r2.x, X[k], fmaf(r2.y, Y[k], fmaf(r2.z, Z[k], r2.w)));
            const float iz = __builtin_amdgcn_rcpf(ez);
            const float fz = f * iz;
            const float du = fmaf(ex, fz, pu[k]);    // u - px,  px = -f ex/ez + cx
            const float dv = fmaf(-ey, fz, pv[k]);   // v - py,  py =  f ey/ez + cy
            const float err = __builtin_amdgcn_sqrtf(fmaf(dv, dv, du * du));
            // guards of the reference: |E.z| < 1e-8 -> 0 ; err > CNN_OBJ_MAXINPUT -> 0
            const bool keep = valid && (fabsf(ez) >= 1e-8f) && !(err > clampv);
            float w;
            if (SOFTMODE) {
                const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kA, fminf(err, clampv), kB)));
                w = s_g[h] * (-beta) * s * (1.0f - s);
            } else {
                w = wv[k];
            }
            w = keep ? w : 0.f;
            const float ie = __builtin_amdgcn_rcpf(err + 1e-8f);
            const float a0 = -du * ie, a1 = -dv * ie;
            const float c0 = -a0 * fz * w;
            const float c1 = a1 * fz * w;
            const float c2 = (a0 * ex - a1 * ey) * fz * iz * w;
            // guard against inf * 0 when ez == 0 (w is already 0 there)
            const float C0 = keep ? c0 : 0.f, C1 = keep ? c1 : 0.f, C2 = keep ? c2 : 0.f;
            gx[k][0] = fmaf(r0.x, C0, fmaf(r1.x, C1, fmaf(r2.x, C2, gx[k][0])));
            gx[k][1] = fmaf(r0.y, C0, fmaf(r1.y, C1, fmaf(r2.y, C2, gx[k][1])));
            gx[k][2] = fmaf(r0.z, C0, fmaf(r1.z, C1, fmaf(r2.z, C2, gx[k][2])));
            G[0] = fmaf(C0, X[k], G[0]); G[1] = fmaf(C0, Y[k], G[1]); G[2] = fmaf(C0, Z[k], G[2]);
            G[3] = fmaf(C1, X[k], G[3]); G[4] = fmaf(C1, Y[k], G[4]); G[5] = fmaf(C1, Z[k], G[5]);
            G[6] = fmaf(C2, X[k], G[6]); G[7] = fmaf(C2, Y[k], G[7]); G[8] = fmaf(C2, Z[k], G[8]);
            G[9] += C0; G[10] += C1; G[11] += C2;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const float s = wave_sum_f(G[i]);
            if (lane == 0) s_red[wave * 12 + i] = s;
        }
        __syncthreads();
        if (tid < 12) {
            float s = 0.f;
#pragma unroll
            for (int wq = 0; wq < K4_THREADS / 64; wq++) s += s_red[wq * 12 + tid];
            G12_part[((size_t)pt * N + h0 + h) * 12 + tid] = s;
        }
        __syncthreads();
    }

    if (valid) {
        float* dst = grad_part + (size_t)ht * P * 3 + (size_t)p0 * 3;
        if (PX == 4) {
            f4* d4 = reinterpret_cast<f4*>(dst);
            d4[0] = f4{gx[0][0], gx[0][1], gx[0][2], gx[1][0]};
            d4[1] = f4{gx[1][1], gx[1][2], gx[2][0], gx[2][1]};
            d4[2] = f4{gx[2][2], gx[3][0], gx[3][1], gx[3][2]};
        } else {
            dst[0] = gx[0][0]; dst[1] = gx[0][1]; dst[2] = gx[0][2];
        }
    }
}

hipError_t score_backward(hipStream_t st, int N, const float* staged_bwd, const FrameDev& F, const float* d_err, const double* g, float clampv,
                          float tau, float beta, float* grad_part, float* G12_part, int* pixel_tiles_used) {
    if (pixel_tiles_used) *pixel_tiles_used = 0;
    if (N <= 0) return hipSuccess;
    const bool soft = d_err == nullptr;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(F.uv) & 15) == 0) && ((reinterpret_cast<uintptr_t>(grad_part) & 15) == 0);
    const int PX = vec ? 4 : 1;
    const int tile = K4_THREADS * PX;
    const int PT = (F.P + tile - 1) / tile;
    const int NT = (N + K4_HT - 1) / K4_HT;
    const int grid = ((PT + 7) / 8) * 8 * NT;
    if (pixel_tiles_used) *pixel_tiles_used = PT;
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool UV = F.uv != nullptr;
#define DSAC_K4(PXV, S, U)                                                                                                               \
    hipLaunchKernelGGL((k_score_backward<PXV, S, U>), dim3(grid), dim3(K4_THREADS), 0, st, staged_bwd, F.xyz, F.uv, d_err, g, grad_part, \
                       G12_part, N, F.P, F.W, PT, NT, F.fx, F.cx, F.cy, clampv, kA, kB, beta)
    if (vec) {
        if (soft) { if (UV) DSAC_K4(4, true, true); else DSAC_K4(4, true, false); }
        else { if (UV) DSAC_K4(4, false, true); else DSAC_K4(4, false, false); }
    } else {
        if (soft) { if (UV) DSAC_K4(1, true, true); else DSAC_K4(1, true, false); }
        else { if (UV) DSAC_K4(1, false, true); else DSAC_K4(1, false, false); }
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

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
#include "dmath.h"

namespace dk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

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

template <int PX, int HT, bool ERR, bool SOFT, bool UV, bool SPOSE>
__global__ __launch_bounds__(K2_THREADS) void k_reproject(const float* __restrict__ staged, const float* __restrict__ xyz,
                                                          const float* __restrict__ uv, float* __restrict__ err,
                                                          float* __restrict__ soft_part, int N, int P, int W, int PT, int NT, float cx,
                                                          float cy, float clampv, float kA, float kB) {
    // XCD-aware decode: the 8 blocks of one dispatch round-robin group cover 8 different pixel tiles, and
    // successive groups walk the hypothesis tiles of those same pixel tiles.
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ht = q % NT;
    const int pt = (q / NT) * 8 + (b & 7);
    if (pt >= PT) return;
    const int h0 = ht * HT;
    const int nh = min(HT, N - h0);
    const int tid = threadIdx.x;

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

    float acc[SOFT ? HT : 1];
    float* erow = ERR ? err + (size_t)h0 * P + p0 : nullptr;

#pragma unroll(SOFT ? HT : 4)
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
#pragma unroll
            for (int k = 0; k < PX; k++) e[k] = residual(r0, r1, r2, X[k], Y[k], Z[k], pu[k], pv[k], clampv);
            if (ERR && valid) {
                if (PX == 4) {
                    f4 o = {e[0], e[1], e[2], e[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f4*>(erow + (size_t)h * P));
                } else {
                    __builtin_nontemporal_store(e[0], erow + (size_t)h * P);
                }
            }
            if (SOFT) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < PX; k++) s += soft_inlier(e[k], kA, kB);
                acc[h] = valid ? s : 0.f;
            }
        } else if (SOFT) {
            acc[h] = 0.f;
        }
    }

    if (SOFT) {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int h = 0; h < HT; h++) {
            const float s = wave_sum(acc[h]);
            if (lane == 0) s_red[wave * HT + h] = s;
        }
        __syncthreads();
        if (tid < nh) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < K2_THREADS / 64; w++) s += s_red[w * HT + tid];
            soft_part[(size_t)pt * N + h0 + tid] = s;
        }
    }
}

int reproject_num_pixel_tiles(int P) { return (P + K2_THREADS - 1) / K2_THREADS; }  // the scalar path's (larger) count

template <int PX, int HT, bool SPOSE>
static hipError_t launch_reproject(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float kA, float kB,
                                   float* soft_part) {
    const int tile = K2_THREADS * PX;
    const int PT = (F.P + tile - 1) / tile;
    const int NT = (N + HT - 1) / HT;
    const int grid = ((PT + 7) / 8) * 8 * NT;
    const bool ERR = err != nullptr, SOFT = soft_part != nullptr, UV = F.uv != nullptr;
#define DSAC_K2(E, S, U)                                                                                                              \
    hipLaunchKernelGGL((k_reproject<PX, HT, E, S, U, SPOSE>), dim3(grid), dim3(K2_THREADS), 0, st, staged, F.xyz, F.uv, err, soft_part, N, \
                       F.P, F.W, PT, NT, F.cx, F.cy, clampv, kA, kB)
    if (ERR && SOFT) { if (UV) DSAC_K2(true, true, true); else DSAC_K2(true, true, false); }
    else if (ERR) { if (UV) DSAC_K2(true, false, true); else DSAC_K2(true, false, false); }
    else if (SOFT) { if (UV) DSAC_K2(false, true, true); else DSAC_K2(false, true, false); }
#undef DSAC_K2
    return hipGetLastError();
}

// variant: 0 = LDS-staged poses, HT = 32 (default) ; 1 = scalar-load poses (SGPR operands), HT = 32 ;
//          2 = LDS, HT = 16 ; 3 = LDS, HT = 64
hipError_t reproject(hipStream_t st, int N, const float* staged, const FrameDev& F, float clampv, float* err, float tau, float beta,
                     float* soft_part, int variant, int* tiles_used) {
    if (tiles_used) *tiles_used = 0;
    if (N <= 0 || F.P <= 0 || (!err && !soft_part)) return hipSuccess;
    const float LOG2E = 1.4426950408889634f;
    const float kA = beta * LOG2E, kB = -beta * tau * LOG2E;
    const bool vec = (F.P % 4 == 0) && ((reinterpret_cast<uintptr_t>(err) & 15) == 0) && ((reinterpret_cast<uintptr_t>(F.xyz) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(F.uv) & 15) == 0);
    if (tiles_used) *tiles_used = vec ? (F.P + K2_THREADS * 4 - 1) / (K2_THREADS * 4) : (F.P + K2_THREADS - 1) / K2_THREADS;
    if (!vec) return launch_reproject<1, 32, false>(st, N, staged, F, clampv, err, kA, kB, soft_part);
    switch (variant) {
        case 1: return launch_reproject<4, 32, true>(st, N, staged, F, clampv, err, kA, kB, soft_part);
        case 2: return launch_reproject<4, 16, false>(st, N, staged, F, clampv, err, kA, kB, soft_part);
        case 3: return launch_reproject<4, 64, false>(st, N, staged, F, clampv, err, kA, kB, soft_part);
        default: return launch_reproject<4, 32, false>(st, N, staged, F, clampv, err, kA, kB, soft_part);
    }
}

// --------------------------------------------------------------------------------------------------
// One wave per hypothesis, lanes stride over the pixel tiles (independent
// loads in flight), fixed butterfly -> deterministic.
__global__ __launch_bounds__(256) void k_reduce_soft(int N, int tiles, const float* __restrict__ part, double* __restrict__ soft) {
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (h >= N) return;
    double s = 0;
    for (int t = lane; t < tiles; t += 64) s += (double)part[(size_t)t * N + h];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) soft[h] = s;
}

hipError_t reduce_soft(hipStream_t st, int N, int tiles, const float* soft_part, double* soft) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_soft, dim3((N + 3) / 4), dim3(256), 0, st, N, tiles, soft_part, soft);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// K3: softmax (max-subtracted), Shannon entropy in bits, weighted pose average.  One workgroup, fp64,
// fixed reduction tree -> deterministic.  Replaces core/cnn_softam.h:535-553, :80-88, :1082-1094.
// --------------------------------------------------------------------------------------------------
constexpr int K3_THREADS = 256;

template <typename Op>
DM_INLINE double block_reduce(double v, double* s_buf, Op op) {
    const int tid = threadIdx.x;
    s_buf[tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = K3_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) s_buf[tid] = op(s_buf[tid], s_buf[tid + o]);
        __syncthreads();
    }
    const double r = s_buf[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(K3_THREADS) void k_softmax(int N, const double* __restrict__ scores, double scale, double* __restrict__ w,
                                                        double* __restrict__ entropy, const double* __restrict__ poses,
                                                        double* __restrict__ avg6) {
    __shared__ double s_buf[K3_THREADS];
    const int tid = threadIdx.x;
    double m = -1.7976931348623157e308;
    for (int i = tid; i < N; i += K3_THREADS) m = fmax(m, scale * scores[i]);
    m = block_reduce(m, s_buf, [](double a, double b) { return fmax(a, b); });
    double sum = 0;
    for (int i = tid; i < N; i += K3_THREADS) {
        const double e = exp(scale * scores[i] - m);
        w[i] = e;
        sum += e;
    }
    sum = block_reduce(sum, s_buf, [](double a, double b) { return a + b; });
    double ent = 0, a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < N; i += K3_THREADS) {
        const double wi = w[i] / sum;
        w[i] = wi;
        if (wi > 0) ent -= wi * log2(wi);
        if (poses && avg6) {
#pragma unroll
            for (int k = 0; k < 6; k++) a[k] += wi * poses[6 * (size_t)i + k];
        }
    }
    if (entropy) {
        ent = block_reduce(ent, s_buf, [](double x, double y) { return x + y; });
        if (tid == 0) *entropy = ent;
    }
    if (poses && avg6) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double s = block_reduce(a[k], s_buf, [](double x, double y) { return x + y; });
            if (tid == 0) avg6[k] = s;
        }
    }
}

hipError_t softmax(hipStream_t st, int N, const double* scores, double scale, double* w, double* entropy, const double* poses, double* avg6) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_softmax, dim3(1), dim3(K3_THREADS), 0, st, N, scores, scale, w, entropy, poses, avg6);
    return hipGetLastError();
}

}  // namespace dk

// Store-pattern microbenchmark: how fast can the N x P error-image be written as a function of the tile shape?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// block (rt, ct): rows [rt*R, rt*R+R), cols [ct*C4*1024, ...) floats; C4 = number of 1024-float (4 KiB) chunks per row per block
template <bool NT, bool ROWINNER>
__global__ __launch_bounds__(256) void k_store(float* out, int N, int P, int R, int C4, int RT, int CT, int order) {
    int b = blockIdx.x;
    int rt, ct;
    if (order == 0) { ct = b % CT; rt = b / CT; }        // column tiles innermost (pixel-minor)
    else if (order == 1) { rt = b % RT; ct = b / RT; }   // row tiles innermost (hyp-minor)
    else { int q = b >> 3; int CTG = (CT + 7) >> 3; rt = q / CTG; ct = (q % CTG) * 8 + (b & 7); if (ct >= CT) return; }  // xcd-aware pixel-minor
    const int tid = threadIdx.x;
    const f4 v = {(float)b, (float)tid, 1.f, 2.f};
    if (ROWINNER) {
        for (int k = 0; k < C4; k++) {
            const size_t col = ((size_t)ct * C4 + k) * 1024 + tid * 4;
            if (col >= (size_t)P) continue;
            for (int r = 0; r < R; r++) {
                const int row = rt * R + r;
                if (row >= N) break;
                f4* p = reinterpret_cast<f4*>(out + (size_t)row * P + col);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        }
    } else {
        for (int r = 0; r < R; r++) {
            const int row = rt * R + r;
            if (row >= N) break;
            for (int k = 0; k < C4; k++) {
                const size_t col = ((size_t)ct * C4 + k) * 1024 + tid * 4;
                if (col >= (size_t)P) continue;
                f4* p = reinterpret_cast<f4*>(out + (size_t)row * P + col);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        }
    }
}

int main() {
    const int N = 4096, P = 307200;
    float* out;
    CK(hipMalloc(&out, (size_t)N * P * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int Rs[] = {2, 4, 8, 1024, 4096};
    const int Cs[] = {1, 2};
    for (int nt = 0; nt < 2; nt++)
    for (int rowinner = 0; rowinner < 2; rowinner++)
    for (int order = 0; order < 3; order++)
    for (int R : Rs) for (int C4 : Cs) {
        if (rowinner == 0 && R > 1 && C4 > 1 && order != 0) continue;
        const int RT = (N + R - 1) / R, CT = (P + C4 * 1024 - 1) / (C4 * 1024);
        const int grid = order == 2 ? ((CT + 7) / 8) * 8 * RT : RT * CT;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            if (nt) { if (rowinner) hipLaunchKernelGGL((k_store<true, true>), dim3(grid), dim3(256), 0, 0, out, N, P, R, C4, RT, CT, order);
                      else hipLaunchKernelGGL((k_store<true, false>), dim3(grid), dim3(256), 0, 0, out, N, P, R, C4, RT, CT, order); }
            else { if (rowinner) hipLaunchKernelGGL((k_store<false, true>), dim3(grid), dim3(256), 0, 0, out, N, P, R, C4, RT, CT, order);
                   else hipLaunchKernelGGL((k_store<false, false>), dim3(grid), dim3(256), 0, 0, out, N, P, R, C4, RT, CT, order); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("nt=%d rowinner=%d order=%d R=%3d C=%2dKiB grid=%7d : %7.1f us  %6.0f GB/s\n", nt, rowinner, order, R, C4 * 4, grid, best * 1e3,
               (double)N * P * 4 / best / 1e6);
    }
    return 0;
}

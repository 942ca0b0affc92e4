// Store-locality microbenchmark (round 3): WHY does the fully sequential pattern of store_pattern.hip (one 4-KiB chunk per workgroup,
// workgroup b -> chunk b) stream at 7.0-7.4 TB/s when every tiled pattern stays at 5.8-6.3?
//   Hypothesis: workgroup b runs on XCD b % 8 and 4-KiB chunk g of the address space is served by memory stack g % 8 -- the sequential pattern
//   is the only one of round 1's table in which every XCD writes ONLY chunks with g % 8 == its own id.
// Modes:
//   A s        : one chunk per workgroup, chunk = 8 * (b / 8) + ((b % 8 + s) % 8)  (s = 0: sequential; s != 0: same addresses in the same
//                time order, but XCD x writes the chunks of class x + s)
//   B R par    : K2-shaped tiles, R rows x one 4-KiB chunk per workgroup, XCD-aware pixel-minor block order (K2's decode).  par = 0: rows
//                rt*R .. rt*R+R-1 at chunk column ct (ct % 8 == xcd) -- the row stride is 300 chunks, 300 % 8 == 4, so odd rows land in class
//                xcd + 4.  par = 1: a tile takes R rows of ONE parity (rt2*2R + 2i + p) at chunk column ct ^ (4p): every store is class xcd.
//   C R par    : as B, with the matrix-core K2's instruction shape (a wave store covers 4 rows x 256 B, 4 chunks of 64 pixels per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_modeA(float* out, int shift, size_t chunks) {
    const size_t b = blockIdx.x;
    const size_t g = (b & ~(size_t)7) | ((b + shift) & 7);
    if (g >= chunks) return;
    const f4 v = {(float)b, 1.f, 2.f, 3.f};
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + g * 1024 + threadIdx.x * 4));
}

// c0: chunk class of the buffer's first chunk ((address >> 12) & 7)
template <bool HP>
__global__ __launch_bounds__(256) void k_modeB(float* out, int N, int P, int R, int par, int CT, int c0) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int CTG = (CT + 7) >> 3;
    const int ctg = q % CTG, rt = q / CTG;
    int row0, rstride, ct;
    if (par) {
        const int p = rt & 1;
        row0 = (rt >> 1) * 2 * R + p;
        rstride = 2;
        ct = ctg * 8 + (((xcd - c0) & 7) ^ (4 * p));
    } else {
        row0 = rt * R;
        rstride = 1;
        ct = ctg * 8 + ((xcd - c0) & 7);
    }
    if (ct >= CT) return;
    const int tid = threadIdx.x;
    const f4 v = {(float)b, (float)tid, 2.f, 3.f};
    if (!HP) {
        for (int r = 0; r < R; r++) {
            const int row = row0 + r * rstride;
            if (row >= N) break;
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)row * P + (size_t)ct * 1024 + tid * 4));
        }
    } else {
        // wave w: chunks of 64 pixels w*4 .. w*4+3; lane (g, c): 4 consecutive pixels 4c of 4 rows 4g .. 4g+3 of every 16-row group
        const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
        for (int gi = 0; gi < R / 16; gi++)
            for (int ch = 0; ch < 4; ch++)
                for (int r = 0; r < 4; r++) {
                    const int row = row0 + (gi * 16 + 4 * g + r) * rstride;
                    if (row >= N) continue;
                    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)row * P + (size_t)ct * 1024 + (wave * 4 + ch) * 64 + 4 * c));
                }
    }
}

int main() {
    const int N = 4096, P = 307200;
    float* out;
    CK(hipMalloc(&out, (size_t)N * P * 4));
    const int c0 = (int)((reinterpret_cast<uintptr_t>(out) >> 12) & 7);
    printf("buffer %p, first chunk class %d\n", (void*)out, c0);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t chunks = (size_t)N * P / 1024;
    auto timeit = [&](auto launch, const char* name) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(a));
            launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0) { if (ms < best) best = ms; sum += ms; }
        }
        printf("%-40s : best %7.1f us  mean %7.1f us  %6.0f GB/s\n", name, best * 1e3, sum / 4 * 1e3, (double)N * P * 4 / best / 1e6);
    };
    char name[128];
    for (int rep = 0; rep < 2; rep++)
    for (int s = 0; s < 8; s++) {
        snprintf(name, sizeof(name), "A shift %d (pass %d)", s, rep);
        timeit([&] { hipLaunchKernelGGL(k_modeA, dim3((unsigned)chunks), dim3(256), 0, 0, out, s, chunks); }, name);
    }
    const int CT = P / 1024;
    const int Rs[] = {16, 32, 64, 128};
    for (int hp = 0; hp < 2; hp++)
    for (int R : Rs)
    for (int par = 0; par < 2; par++) {
        const int RT = (N + R - 1) / R;
        const int grid = ((CT + 7) / 8) * 8 * RT;
        snprintf(name, sizeof(name), "%s R=%3d parity-split %d", hp ? "C (hp store shape)" : "B (row stores)", R, par);
        if (hp) timeit([&] { hipLaunchKernelGGL((k_modeB<true>), dim3(grid), dim3(256), 0, 0, out, N, P, R, par, CT, c0); }, name);
        else timeit([&] { hipLaunchKernelGGL((k_modeB<false>), dim3(grid), dim3(256), 0, 0, out, N, P, R, par, CT, c0); }, name);
    }
    // wrong class on purpose: parity split with the class offset by 1..4 (everything remote)
    for (int off = 1; off <= 4; off++) {
        const int R = 64, RT = (N + R - 1) / R, grid = ((CT + 7) / 8) * 8 * RT;
        snprintf(name, sizeof(name), "B R= 64 parity-split, class offset %d", off);
        timeit([&] { hipLaunchKernelGGL((k_modeB<false>), dim3(grid), dim3(256), 0, 0, out, N, P, R, 1, CT, (c0 + off) & 7); }, name);
    }
    return 0;
}

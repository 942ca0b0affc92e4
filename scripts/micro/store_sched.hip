// Store-schedule microbenchmark (round 3): which K2-shaped write schedule streams fastest, and what decides it?
// Every kernel writes the full N x P fp32 error-image volume with the matrix-core K2's store instruction (a wave store covers 4 rows x 256 B:
// lane (g, c) writes 4 consecutive pixels 4c of rows 4g + r of a 16-row group).  Parameters:
//   R      rows (hypotheses) per workgroup tile: 16 / 32 / 64
//   CHW    64-pixel chunks per wave: 1 / 2 / 4
//   WAVES  waves per workgroup: 1 / 4 / 8 / 16     (pixel tile of a workgroup = WAVES * CHW * 64 pixels)
//   order  0 plain pixel-minor, 2 XCD-aware pixel-minor (K2's decode), 3 XCD-contiguous (XCD x owns the x-th eighth of every row)
//   work   dummy dependent VALU iterations before the stores (a wave's arithmetic: how long it lives before it stores)
//   lds    bytes of unused LDS per workgroup (caps the resident workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_tile(float* out, int N, int P, int R, int CHW, int order, int work, float seedv) {
    extern __shared__ float s_pad[];
    const int waves = blockDim.x >> 6;
    const int tile_px = waves * CHW * 64;
    const int PT = (P + tile_px - 1) / tile_px;
    const int b = blockIdx.x;
    int rt, pt;
    if (order == 0) { pt = b % PT; rt = b / PT; }
    else if (order == 2) { const int q = b >> 3, PTG = (PT + 7) >> 3; rt = q / PTG; pt = (q % PTG) * 8 + (b & 7); }
    else { const int q = b >> 3, PTG = (PT + 7) >> 3; rt = q / PTG; pt = (b & 7) * PTG + (q % PTG); }
    if (pt >= PT) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float acc = seedv + (float)tid;
    for (int i = 0; i < work; i++) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);  // dependent chain, 1 VALU op per iteration
    if (work < 0) s_pad[tid] = acc;  // never
    const f4 v = {acc, (float)b, 2.f, 3.f};
    for (int gi = 0; gi < R / 16; gi++)
        for (int ch = 0; ch < CHW; ch++) {
            const int col = pt * tile_px + (wave * CHW + ch) * 64 + 4 * c;
            if (col >= P) continue;
            for (int r = 0; r < 4; r++) {
                const int row = rt * R + gi * 16 + 4 * g + r;
                if (row < N) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)row * P + col));
            }
        }
}

// k_tile in plain order with a row stride of its own (Pp floats between the rows, P pixels written per row)
__global__ void k_tile_stride(float* out, int N, int P, int Pp, int R, int CHW, int work, float seedv) {
    const int waves = blockDim.x >> 6;
    const int tile_px = waves * CHW * 64;
    const int PT = (P + tile_px - 1) / tile_px;
    const int b = blockIdx.x;
    const int pt = b % PT, rt = b / PT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float acc = seedv + (float)tid;
    for (int i = 0; i < work; i++) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    const f4 v = {acc, (float)b, 2.f, 3.f};
    for (int gi = 0; gi < R / 16; gi++)
        for (int ch = 0; ch < CHW; ch++) {
            const int col = pt * tile_px + (wave * CHW + ch) * 64 + 4 * c;
            if (col >= P) continue;
            for (int r = 0; r < 4; r++) {
                const int row = rt * R + gi * 16 + 4 * g + r;
                if (row < N) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)row * Pp + col));
            }
        }
}

// the same tiles with ROW-CONTIGUOUS store instructions: a wave store covers ONE row x 1 KiB (lane l writes pixels 4l .. 4l+3 of a 256-pixel span) instead
// of 4 rows x 256 B -- what an LDS transpose of K2's tile would give.  tile_px must be a multiple of 256; the R x (tile_px / 256) spans of the tile are
// dealt round-robin to the waves.
__global__ void k_tile_rows(float* out, int N, int P, int R, int CHW, int order, int work, float seedv) {
    const int waves = blockDim.x >> 6;
    const int tile_px = waves * CHW * 64;
    const int PT = (P + tile_px - 1) / tile_px;
    const int b = blockIdx.x;
    int rt, pt;
    if (order == 0) { pt = b % PT; rt = b / PT; }
    else { const int q = b >> 3, PTG = (PT + 7) >> 3; rt = q / PTG; pt = (q % PTG) * 8 + (b & 7); }
    if (pt >= PT) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = seedv + (float)tid;
    for (int i = 0; i < work; i++) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    const f4 v = {acc, (float)b, 2.f, 3.f};
    const int spans = tile_px / 256;
    for (int i = wave; i < R * spans; i += waves) {
        const int row = rt * R + i / spans, col = pt * tile_px + (i % spans) * 256 + 4 * lane;
        if (row < N && col < P) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)row * P + col));
    }
}

// one row x 4 KiB per workgroup, sequential (the fastest pattern of round 1's table) with the same dummy work
__global__ __launch_bounds__(256) void k_seq(float* out, size_t chunks, int work, float seedv) {
    const size_t b = blockIdx.x;
    float acc = seedv + (float)threadIdx.x;
    for (int i = 0; i < work; i++) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    const f4 v = {acc, 1.f, 2.f, 3.f};
    if (b < chunks) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + b * 1024 + threadIdx.x * 4));
}

// lock-step sweep: every wave owns PXW pixels (a multiple of 4, <= 64 per chunk x CHW chunks) of the row and walks ALL rows in groups of 16 -- the whole grid
// (sized to be co-resident) writes one 16-row band across the full width at a time
__global__ void k_sweep(float* out, int N, int P, int px_per_wave, int work, float seedv) {
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    const int chunks = (px_per_wave + 63) / 64;
    float acc = seedv + (float)tid;
    for (int rg = 0; rg < N / 16; rg++) {
        for (int i = 0; i < work; i++) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
        const f4 v = {acc, (float)rg, 2.f, 3.f};
        for (int ch = 0; ch < chunks; ch++) {
            const int off = ch * 64 + 4 * c;
            const int col = gw * px_per_wave + off;
            if (off >= px_per_wave || col >= P) continue;
            for (int r = 0; r < 4; r++)
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)(rg * 16 + 4 * g + r) * P + col));
        }
    }
}

int main(int argc, char** argv) {
    const int N = 4096, P = 307200;
    float* out;
    CK(hipMalloc(&out, (size_t)N * P * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto timeit = [&](auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        return best;
    };
    const size_t chunks = (size_t)N * P / 1024;
    const bool rows_only = argc > 1 && (argv[1][0] == 'r' || argv[1][0] == 'p');
    for (int work : {0, 200, 800}) {
        if (rows_only) break;
        const float ms = timeit([&] { hipLaunchKernelGGL(k_seq, dim3((unsigned)chunks), dim3(256), 0, 0, out, chunks, work, 1.f); });
        printf("seq 1 row x 4 KiB per workgroup, work %4d : %7.1f us  %6.0f GB/s\n", work, ms * 1e3, (double)N * P * 4 / ms / 1e6);
    }
    for (int work : {0, 100, 300})
        for (int ppw : {60, 64, 120, 128, 240, 256, 300}) if (!rows_only)
            for (int waves : {1, 4}) {
                const int nw = (P + ppw - 1) / ppw, grid = (nw + waves - 1) / waves;
                const float ms = timeit([&] { hipLaunchKernelGGL(k_sweep, dim3(grid), dim3(waves * 64), 0, 0, out, N, P, ppw, work, 1.f); });
                printf("sweep px/wave %3d waves/wg %d (%5d waves) work %3d : %7.1f us  %6.0f GB/s\n", ppw, waves, nw, work, ms * 1e3, (double)N * P * 4 / ms / 1e6);
            }
    if (argc > 1 && argv[1][0] == 'p') {  // row stride: does the distance between the rows of a tile (P * 4 bytes = 300 x 4 KiB for 640 x 480) matter?
        struct C2 { int R, CHW, WAVES; };
        float* big;
        CK(hipMalloc(&big, (size_t)N * (P + 8192) * 4));
        for (int work : {0, 300})
            for (C2 c : {C2{32, 4, 1}, C2{64, 4, 4}, C2{16, 1, 4}})
                for (int pad : {0, 64, 256, 320, 1024, 1088, 4096, 4160, 8192}) {
                    const int Pp = P + pad;  // the tiles cover the first P pixels of every row, rows are Pp floats apart
                    const int tile_px = c.WAVES * c.CHW * 64;
                    const int PT = (P + tile_px - 1) / tile_px, RT = (N + c.R - 1) / c.R;
                    const float ms = timeit([&] { hipLaunchKernelGGL(k_tile_stride, dim3((unsigned)(PT * RT)), dim3(c.WAVES * 64), 0, 0, big, N, P, Pp, c.R, c.CHW, work, 1.f); });
                    printf("R %2d CHW %d WAVES %2d work %3d row stride %7d floats (+%4d) : %7.1f us  %6.0f GB/s\n", c.R, c.CHW, c.WAVES, work, Pp, pad, ms * 1e3,
                           (double)N * P * 4 / ms / 1e6);
                }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'r') {  // store-instruction shape A/B on K2's tiles
        struct C2 { int R, CHW, WAVES; };
        for (int work : {0, 300})
            for (C2 c : {C2{32, 4, 1}, C2{16, 4, 1}, C2{64, 4, 1}, C2{64, 1, 4}, C2{16, 1, 4}, C2{32, 1, 4}, C2{16, 4, 4}, C2{64, 4, 4}})
                for (int order : {0, 2})
                    for (int shape : {0, 1}) {
                        const int tile_px = c.WAVES * c.CHW * 64;
                        const int PT = (P + tile_px - 1) / tile_px, RT = (N + c.R - 1) / c.R;
                        const long long grid = order == 0 ? (long long)PT * RT : (long long)((PT + 7) / 8) * 8 * RT;
                        const float ms = timeit([&] {
                            if (shape == 0) hipLaunchKernelGGL(k_tile, dim3((unsigned)grid), dim3(c.WAVES * 64), 0, 0, out, N, P, c.R, c.CHW, order, work, 1.f);
                            else hipLaunchKernelGGL(k_tile_rows, dim3((unsigned)grid), dim3(c.WAVES * 64), 0, 0, out, N, P, c.R, c.CHW, order, work, 1.f);
                        });
                        printf("R %2d CHW %d WAVES %2d order %d work %3d store = %s : %7.1f us  %6.0f GB/s\n", c.R, c.CHW, c.WAVES, order, work,
                               shape ? "1 row x 1 KiB " : "4 rows x 256 B", ms * 1e3, (double)N * P * 4 / ms / 1e6);
                    }
        return 0;
    }
    if (argc > 1) return 0;
    struct Cfg { int R, CHW, WAVES, order, work, lds; };
    std::vector<Cfg> cfgs;
    for (int work : {0, 300})
        for (int R : {16, 32, 64})
            for (int CHW : {1, 2, 4})
                for (int WAVES : {1, 4, 16})
                    for (int order : {0, 2, 3})
                        cfgs.push_back({R, CHW, WAVES, order, work, 0});
    // occupancy caps on the interesting shapes (LDS per workgroup: 160 KiB / lds = resident workgroups per CU)
    for (int work : {0, 300})
        for (int lds : {20 * 1024, 40 * 1024, 80 * 1024})
            for (Cfg base : {Cfg{16, 1, 4, 2, 0, 0}, Cfg{16, 1, 16, 2, 0, 0}, Cfg{64, 4, 4, 2, 0, 0}, Cfg{64, 1, 4, 2, 0, 0}, Cfg{32, 1, 16, 2, 0, 0}, Cfg{16, 4, 4, 2, 0, 0}})
                cfgs.push_back({base.R, base.CHW, base.WAVES, base.order, work, lds});
    for (const Cfg& c : cfgs) {
        const int tile_px = c.WAVES * c.CHW * 64;
        const int PT = (P + tile_px - 1) / tile_px, RT = (N + c.R - 1) / c.R;
        const long long grid = c.order == 0 ? (long long)PT * RT : (long long)((PT + 7) / 8) * 8 * RT;
        const float ms = timeit([&] { hipLaunchKernelGGL(k_tile, dim3((unsigned)grid), dim3(c.WAVES * 64), c.lds, 0, out, N, P, c.R, c.CHW, c.order, c.work, 1.f); });
        printf("R %2d CHW %d WAVES %2d order %d work %3d lds %6d grid %8lld : %7.1f us  %6.0f GB/s\n", c.R, c.CHW, c.WAVES, c.order, c.work, c.lds, grid, ms * 1e3,
               (double)N * P * 4 / ms / 1e6);
    }
    return 0;
}

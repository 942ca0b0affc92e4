// VALU issue-rate microbenchmark (round 3): what does one wave64 fp32 instruction cost on a gfx950 SIMD -- plain, packed, packed with a
// broadcast half, transcendental -- with 1 / 2 / 4 waves per SIMD, independent accumulators or one dependent chain?  K4's main pass stays at
// ~120 us whatever its instruction count; this prices its instruction mix directly.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_rate.hip -o scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t ev_ = (x); if (ev_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(ev_)); exit(1); } } while (0)

enum { K_FMA, K_PKFMA, K_PKMUL, K_PKADD, K_PKFMA_BCAST, K_PKFMA_SGPR, K_MUL, K_RSQ, K_PKFMA_DEP, K_FMA_DEP, K_MIX_PK_FMA, K_PKFMA_ILP4, K_PKFMA_ILP2,
       K_FMA_ILP4, K_PKFMA_MFMA, K_FMA_MFMA, K_PKFMA_3SRC, K_MFMA_F32, K_MFMA_BF16, K_PKFMA_MFMA_BF16, K_PKFMA4_MFMA_BF16, K_RSQ_MFMA_BF16, K_SQRT, K_RCP, K_EXP, K_MIN, K_RSQ_PK, K_NKINDS };
static const char* names[] = {"v_fma_f32 x16 indep", "v_pk_fma_f32 x16 indep", "v_pk_mul_f32 x16 indep", "v_pk_add_f32 x16 indep",
                              "v_pk_fma_f32 bcast(op_sel_hi 0) x16", "v_pk_fma_f32 sgpr src0 x16", "v_mul_f32 x16 indep", "v_rsq_f32 x16 indep",
                              "v_pk_fma_f32 dependent chain", "v_fma_f32 dependent chain", "pk_fma + fma alternating x16", "v_pk_fma_f32 ILP 4",
                              "v_pk_fma_f32 ILP 2", "v_fma_f32 ILP 4", "8 pk_fma per mfma16x16x4 (x16 indep)", "16 fma per mfma16x16x4",
                              "v_pk_fma_f32 3 distinct vgpr srcs x16", "v_mfma_f32_16x16x4_f32 alone (4 accumulators)",
                              "v_mfma_f32_16x16x32_bf16 alone (4 accumulators)", "8 pk_fma per mfma16x16x32_bf16", "4 pk_fma per mfma16x16x32_bf16",
                              "4 rsq per mfma16x16x32_bf16", "v_sqrt_f32 x16 indep", "v_rcp_f32 x16 indep", "v_exp_f32 x16 indep", "v_min_f32 x16 indep",
                              "v_rsq_f32 + v_pk_fma_f32 alternating"};

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, int iters, float s) {
    f2 a[16];
    float q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = f2{s + (float)i + (float)threadIdx.x, s - (float)i}; q[i] = s + (float)i * 0.5f + (float)threadIdx.x; }
    f2 b = {1.0000001f + s, 0.9999999f + s}, c = {1e-9f + s, -1e-9f + s}, d = {0.5f + s, 0.25f + s};
    float bs = 1.0000001f + s, cs = 1e-9f + s;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 accs[4] = {acc, acc, acc, acc};
    f4 a4 = {s, s + 1.f, s, s}, b4 = {s + 2.f, s, s, s + 3.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(bs), "v"(cs));
                if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == K_PKFMA_BCAST) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == K_PKFMA_SGPR) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(b), "v"(c));
                if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(bs));
                if (KIND == K_RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]));
                if (KIND == K_PKFMA_DEP) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                if (KIND == K_FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[0]) : "v"(bs), "v"(cs));
                if (KIND == K_MIX_PK_FMA) {
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(bs), "v"(cs));
                }
                if (KIND == K_PKFMA_ILP4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(b), "v"(c));
                if (KIND == K_PKFMA_ILP2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(b), "v"(c));
                if (KIND == K_FMA_ILP4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i & 3]) : "v"(bs), "v"(cs));
                if (KIND == K_PKFMA_MFMA) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    if ((i & 7) == 7) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(bs), "v"(cs));
                }
                if (KIND == K_FMA_MFMA) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(bs), "v"(cs));
                    if (i == 15) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(bs), "v"(cs));
                }
                if (KIND == K_MFMA_F32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accs[i & 3]) : "v"(bs), "v"(cs));
                if (KIND == K_MFMA_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accs[i & 3]) : "v"(a4), "v"(b4));
                if (KIND == K_PKFMA_MFMA_BF16) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    if ((i & 7) == 7) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accs[(i >> 3) + 2 * r]) : "v"(a4), "v"(b4));
                }
                if (KIND == K_PKFMA4_MFMA_BF16) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    if ((i & 3) == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accs[(i >> 2) & 3]) : "v"(a4), "v"(b4));
                }
                if (KIND == K_RSQ_MFMA_BF16) {
                    asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]));
                    if ((i & 3) == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accs[(i >> 2) & 3]) : "v"(a4), "v"(b4));
                }
                if (KIND == K_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(q[i]));
                if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(q[i]));
                if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(q[i]));
                if (KIND == K_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(q[i]) : "v"(bs));
                if (KIND == K_RSQ_PK) {
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]));
                }
                if (KIND == K_PKFMA_3SRC) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(d));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = acc.x + acc.y + acc.z + acc.w;
    for (int i = 0; i < 4; i++) sum += accs[i].x + accs[i].y + accs[i].z + accs[i].w;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += a[i].x + a[i].y + q[i];
    if (sum == 1234.5678f) out[threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(int wg_per_cu, int iters, float* out, long long* cyc) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, iters / 10, 0.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 0.f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ninst = (double)iters * 32.0;
    const double med = (double)h[h.size() / 2];
    // per SIMD: wg_per_cu waves (one wave of each workgroup per SIMD), each ninst instructions
    printf("%-42s waves/SIMD %d: kernel %8.1f us | per wave: %6.2f ticks/instr (median; min %.2f max %.2f) | per SIMD %6.2f ticks/instr, %6.2f cyc/instr at 2.4 GHz by wall\n",
           names[KIND], wg_per_cu, ms * 1e3, med / ninst, (double)h.front() / ninst, (double)h.back() / ninst, med / ninst / wg_per_cu,
           ms * 1e-3 * 2.4e9 / (ninst * wg_per_cu));
    fflush(stdout);
}


// K2's instruction mix per 16 hypotheses x 16 pixels (one m of hp_chunk): 3 exact-fp32 MFMAs, 16 transcendentals, 15 packed, 14 plain -- in three orders
//   0: MFMAs, then transcendentals, then the rest (what the compiler emits today)      1: every MFMA followed by a third of the transcendentals
//   2: MFMAs spread among the packed instructions, transcendentals last
template <int ORDER>
__global__ __launch_bounds__(256) void k_mix(float* out, long long* cyc, int iters, float s) {
    f2 a[16]; float q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = f2{s + (float)i + (float)threadIdx.x, s - (float)i}; q[i] = s + (float)i * 0.5f + (float)threadIdx.x + 1.f; }
    f2 b = {1.0000001f + s, 0.9999999f + s}, c = {1e-9f + s, -1e-9f + s};
    float bs = 1.0000001f + s, cs = 1e-9f + s;
    f4 acc[3] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#define MF(j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(bs), "v"(cs))
#define TR(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]))
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c))
#define PL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(bs))
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (ORDER == 0) {
            MF(0); MF(1); MF(2);
#pragma unroll
            for (int i = 0; i < 16; i++) TR(i);
#pragma unroll
            for (int i = 0; i < 15; i++) PK(i);
#pragma unroll
            for (int i = 0; i < 14; i++) PL(i);
        } else if (ORDER == 1) {
            MF(0); TR(0); TR(1); TR(2); TR(3); TR(4);
            MF(1); TR(5); TR(6); TR(7); TR(8); TR(9);
            MF(2); TR(10); TR(11); TR(12); TR(13); TR(14); TR(15);
#pragma unroll
            for (int i = 0; i < 15; i++) PK(i);
#pragma unroll
            for (int i = 0; i < 14; i++) PL(i);
        } else {
            PK(0); PK(1); MF(0); PK(2); PK(3); PK(4); PK(5); PK(6); MF(1); PK(7); PK(8); PK(9); PK(10); PK(11); MF(2); PK(12); PK(13); PK(14);
#pragma unroll
            for (int i = 0; i < 14; i++) PL(i);
#pragma unroll
            for (int i = 0; i < 16; i++) TR(i);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int j = 0; j < 3; j++) sum += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += a[i].x + a[i].y + q[i];
    if (sum == 1234.5678f) out[threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ORDER>
static void run_mix(int wg_per_cu, int iters, float* out, long long* cyc) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix<ORDER>, dim3(grid), dim3(256), 0, 0, out, cyc, iters / 10, 0.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mix<ORDER>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 0.f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static const char* nm[] = {"MFMAs, transcendentals, rest", "each MFMA followed by 5-6 transcendentals", "MFMAs among the packed instructions"};
    printf("K2 mix (3 MFMA f32 + 16 rsq + 15 pk_fma + 14 mul), order %d (%s), waves/SIMD %d: %7.1f cycles per block and SIMD at 2.4 GHz by wall (sum of parts: 3x38 + 16x8.8 + 15x5.3 + 14x2.9 = 375)\n",
           ORDER, nm[ORDER], wg_per_cu, ms * 1e-3 * 2.4e9 / ((double)iters * wg_per_cu));
    fflush(stdout);
}

template <int KIND>
static void sweep(int iters, float* out, long long* cyc) {
    for (int w : {1, 2, 4}) run<KIND>(w, iters, out, cyc);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out; long long* cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(long long)));
    if (argc > 2 && argv[2][0] == 'x') {  // K2's mix in three orders
        for (int w : {1, 2, 4, 5}) { run_mix<0>(w, iters, out, cyc); run_mix<1>(w, iters, out, cyc); run_mix<2>(w, iters, out, cyc); }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 't') {  // third round: the transcendental unit
        sweep<K_RSQ>(iters, out, cyc);
        sweep<K_SQRT>(iters, out, cyc);
        sweep<K_RCP>(iters, out, cyc);
        sweep<K_EXP>(iters, out, cyc);
        sweep<K_MIN>(iters, out, cyc);
        sweep<K_RSQ_PK>(iters, out, cyc);
        return 0;
    }
    if (argc > 2) {  // second round: the matrix pipe beside the VALU
        sweep<K_MFMA_F32>(iters, out, cyc);
        sweep<K_MFMA_BF16>(iters, out, cyc);
        sweep<K_PKFMA_MFMA>(iters, out, cyc);
        sweep<K_PKFMA_MFMA_BF16>(iters, out, cyc);
        sweep<K_PKFMA4_MFMA_BF16>(iters, out, cyc);
        sweep<K_RSQ_MFMA_BF16>(iters, out, cyc);
        return 0;
    }
    sweep<K_FMA>(iters, out, cyc);
    sweep<K_MUL>(iters, out, cyc);
    sweep<K_PKFMA>(iters, out, cyc);
    sweep<K_PKFMA_3SRC>(iters, out, cyc);
    sweep<K_PKMUL>(iters, out, cyc);
    sweep<K_PKADD>(iters, out, cyc);
    sweep<K_PKFMA_BCAST>(iters, out, cyc);
    sweep<K_PKFMA_SGPR>(iters, out, cyc);
    sweep<K_RSQ>(iters, out, cyc);
    sweep<K_MIX_PK_FMA>(iters, out, cyc);
    sweep<K_PKFMA_DEP>(iters, out, cyc);
    sweep<K_FMA_DEP>(iters, out, cyc);
    sweep<K_PKFMA_ILP4>(iters, out, cyc);
    sweep<K_PKFMA_ILP2>(iters, out, cyc);
    sweep<K_FMA_ILP4>(iters, out, cyc);
    sweep<K_PKFMA_MFMA>(iters, out, cyc);
    sweep<K_FMA_MFMA>(iters, out, cyc);
    return 0;
}

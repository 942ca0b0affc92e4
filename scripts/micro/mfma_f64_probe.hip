// v_mfma_f64_16x16x4_f64 on gfx950 (round 6): operand / result layout, issue cost alone and inside K2's instruction mix.
// Question behind it: can K2 evaluate E = R.X + t on the fp64 matrix core (exact products, one rounding to float per
// component) and still stay above 0.60 of the HBM roof?  Per 16 hypotheses x 16 pixels the exact form issues 3 fp64 MFMAs + 12 v_cvt_f32_f64
// (+ 1 v_cvt_f64_f32 for the B operand) where the fast form issues 3 exact-fp32 MFMAs.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_f64_probe.hip -o scripts/micro/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t ev_ = (x); if (ev_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(ev_)); exit(1); } } while (0)

// ---- layout ----------------------------------------------------------------------------------------------------------------------------
__global__ void k_layout(const double* a, const double* b, double* d) {
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[threadIdx.x * 4 + r] = c[r];
}

static void layout() {
    // hypothesis: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16].  With A[i][k] = 2^i (k == K0), B[k][j] = 3^... use primes:
    // A[i][k] = (i + 1) + 100 (k + 1), B[k][j] = (j + 1) + 1000 (k + 1): D[i][j] = sum_k A[i][k] B[k][j], unique per (i, j) -> invert by search.
    std::vector<double> A(64), B(64), D(256), ref(256);
    for (int l = 0; l < 64; l++) { const int i = l % 16, k = l / 16; A[l] = (i + 1) + 100.0 * (k + 1); B[l] = (i + 1) * 7.0 + 1000.0 * (k + 1); }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += ((i + 1) + 100.0 * (k + 1)) * ((j + 1) * 7.0 + 1000.0 * (k + 1));
            ref[i * 16 + j] = s;
        }
    double *da, *db, *dd;
    CK(hipMalloc(&da, 64 * 8)); CK(hipMalloc(&db, 64 * 8)); CK(hipMalloc(&dd, 256 * 8));
    CK(hipMemcpy(da, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(D.data(), dd, 256 * 8, hipMemcpyDeviceToHost));
    int blocked = 0, interleaved = 0, unknown = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            int fi = -1, fj = -1;
            for (int i = 0; i < 16 && fi < 0; i++)
                for (int j = 0; j < 16; j++)
                    if (ref[i * 16 + j] == D[l * 4 + r]) { fi = i; fj = j; break; }
            if (fi < 0) { unknown++; continue; }
            if (l < 20 || l == 63) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, fi, fj);
            if (fj == l % 16 && fi == 4 * (l / 16) + r) blocked++;
            if (fj == l % 16 && fi == (l / 16) + 4 * r) interleaved++;
        }
    printf("layout: A lane l = A[l%%16][l/16], B lane l = B[l/16][l%%16] assumed; D entries matching row = 4(l/16)+r: %d, row = (l/16)+4r: %d, unmatched: %d (of 256)\n",
           blocked, interleaved, unknown);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dd));
}


// ---- numerics of the fp16 matrix core's accumulation (round 6) ------------------------------------------------------------------------------
// D[0][0] = C + sum_k a_k b_k with the operands of row 0 / column 0 in lane 0 (k = 0..7), everything else zero.  What the exact-transform K2 wants to know:
// does the instruction round ONCE (wide internal sum) or after every product, to nearest or by truncation, and how far below the largest addend does it keep bits?
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__global__ void k_num(const float* av, const float* bv, float c0, float* out, int k16) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (threadIdx.x == 0) ? (_Float16)av[i] : (_Float16)0.f; b[i] = (threadIdx.x == 0) ? (_Float16)bv[i] : (_Float16)0.f; }
    f4 c = {(threadIdx.x == 0) ? c0 : 0.f, 0.f, 0.f, 0.f};
    f4 d;
    if (k16) d = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), c, 0, 0, 0);
    else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = d.x;
}
static float run_num(const float (&a)[8], const float (&b)[8], float c, int k16 = 0) {
    float *da, *db, *dout, r;
    CK(hipMalloc(&da, 32)); CK(hipMalloc(&db, 32)); CK(hipMalloc(&dout, 4));
    CK(hipMemcpy(da, a, 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_num, dim3(1), dim3(64), 0, 0, da, db, c, dout, k16);
    CK(hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost));
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
    return r;
}
static void numerics() {
    const float big = 16384.f, ulp = 0.001953125f;  // 2^14 and its fp32 ulp 2^-9
    for (int k16 = 0; k16 < 2; k16++) {
        printf("numerics of v_mfma_f32_16x16x%s_f16 (results in ulps of 2^14 above 2^14):\n", k16 ? "16" : "32");
        { const float a[8] = {128.f, 0.046875f, 0, 0, 0, 0, 0, 0}, b[8] = {128.f, 0.03125f, 0, 0, 0, 0, 0, 0};  // 2^14 + 0.75 ulp
          printf("  2^14 + 0.75 ulp (two products)                      -> %+.3f   (nearest: +1, truncation: 0)\n", (run_num(a, b, 0.f, k16) - big) / ulp); }
        { const float a[8] = {128.f, 0.046875f, 0.046875f, 0.046875f, 0, 0, 0, 0}, b[8] = {128.f, 0.015625f, 0.015625f, 0.015625f, 0, 0, 0, 0};  // + 3 x 0.375 ulp
          printf("  2^14 + 3 x 0.375 ulp (four products)                -> %+.3f   (one rounding of the exact sum: +1; rounding per product: 0)\n", (run_num(a, b, 0.f, k16) - big) / ulp); }
        { const float a[8] = {0.046875f, 0.046875f, 0.046875f, 0, 0, 0, 0, 0}, b[8] = {0.015625f, 0.015625f, 0.015625f, 0, 0, 0, 0, 0};
          printf("  C = 2^14, 3 x 0.375 ulp as products                 -> %+.3f   (exact sum then one rounding: +1)\n", (run_num(a, b, big, k16) - big) / ulp); }
        { const float a[8] = {128.f, -0.046875f, -0.046875f, -0.046875f, 0, 0, 0, 0}, b[8] = {128.f, 0.015625f, 0.015625f, 0.015625f, 0, 0, 0, 0};
          printf("  2^14 - 3 x 0.375 ulp                                -> %+.3f   (one rounding: -1; truncation toward zero: -2 (2^14 - 1.125 ulp -> ulp below is half: -2.25 half-ulps))\n", (run_num(a, b, 0.f, k16) - big) / (ulp * 0.5f)); }
        for (int k = 8; k <= 40; k += 2) {  // 2^14 - 2^14 + 2^-k: how many bits below the largest addend survive the alignment?
            const float sm = ldexpf(1.f, -(k / 2)), sm2 = ldexpf(1.f, -(k - k / 2));
            if (sm2 < 6.2e-5f) break;  // keep both factors normal halves
            const float a[8] = {128.f, -128.f, sm, 0, 0, 0, 0, 0}, b[8] = {128.f, 128.f, sm2, 0, 0, 0, 0, 0};
            const float a2[8] = {128.f, sm, -128.f, 0, 0, 0, 0, 0}, b2[8] = {128.f, sm2, 128.f, 0, 0, 0, 0, 0};
            printf("  2^14 - 2^14 + 2^-%d -> %g (exact %g); order (big, small, -big) -> %g\n", k, run_num(a, b, 0.f, k16), ldexpf(1.f, -k), run_num(a2, b2, 0.f, k16));
        }
    }
}

// ---- rates -----------------------------------------------------------------------------------------------------------------------------
enum { R_MFMA64, R_MFMA32, R_CVT_F32_F64, R_CVT_F64_F32, R_FMA64, R_MFMA64_DEP, R_N };
static const char* rnames[] = {"v_mfma_f64_16x16x4_f64 alone (4 accumulators)", "v_mfma_f32_16x16x4_f32 alone (4 accumulators)", "v_cvt_f32_f64 x16 indep",
                               "v_cvt_f64_f32 x16 indep", "v_fma_f64 x16 indep", "v_mfma_f64_16x16x4_f64 dependent accumulator"};

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float s) {
    double qd[16]; float q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { q[i] = s + (float)i * 0.5f + (float)threadIdx.x; qd[i] = (double)q[i] + 0.25; }
    double bd = 1.0000001 + s, cd = 1e-9 + s;
    float bs = 1.0000001f + s, cs = 1e-9f + s;
    d4 dacc[4]; f4 facc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { dacc[i] = d4{0, 0, 0, 0}; facc[i] = f4{0, 0, 0, 0}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == R_MFMA64) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(dacc[i & 3]) : "v"(bd), "v"(cd));
                if (KIND == R_MFMA64_DEP) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(dacc[0]) : "v"(bd), "v"(cd));
                if (KIND == R_MFMA32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(facc[i & 3]) : "v"(bs), "v"(cs));
                if (KIND == R_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(q[i]) : "v"(qd[i]));
                if (KIND == R_CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(qd[i]) : "v"(q[i]));
                if (KIND == R_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(qd[i]) : "v"(bd), "v"(cd));
            }
        }
    }
    double sum = 0;
    for (int i = 0; i < 4; i++) sum += dacc[i].x + dacc[i].y + dacc[i].z + dacc[i].w + facc[i].x + facc[i].y + facc[i].z + facc[i].w;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += qd[i] + q[i];
    if (sum == 1234.5678) out[threadIdx.x] = (float)sum;
}

template <int KIND>
static void run(int wg_per_cu, int iters, float* out) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, iters / 10, 0.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 0.f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ninst = (double)iters * 32.0;
    printf("%-50s waves/SIMD %d: kernel %8.1f us | %6.2f cyc/instr per SIMD at 2.4 GHz by wall\n", rnames[KIND], wg_per_cu, ms * 1e3,
           ms * 1e-3 * 2.4e9 / (ninst * wg_per_cu));
    fflush(stdout);
}

// K2's mix per 16 hypotheses x 16 pixels (one m of hp_chunk): the fast form (3 fp32 MFMAs + 16 transcendentals + 15 packed + 14 plain) against the exact
// form (3 fp64 MFMAs + 12 v_cvt_f32_f64 + 1 v_cvt_f64_f32 + the same VALU tail), the conversions straight behind the MFMA they read (what the data flow
// forces) -- and the exact form with the NEXT m's MFMAs issued before this m's tail (software pipelining: the matrix pipe works under the tail)
template <int FORM>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float s) {
    f2 a[16]; float q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = f2{s + (float)i + (float)threadIdx.x, s - (float)i}; q[i] = s + (float)i * 0.5f + (float)threadIdx.x + 1.f; }
    f2 b = {1.0000001f + s, 0.9999999f + s}, c = {1e-9f + s, -1e-9f + s};
    float bs = 1.0000001f + s, cs = 1e-9f + s;
    double bd = 1.0000001 + s, cd = 1e-9 + s;
    f4 acc[3] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    d4 dac[2][3];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 3; j++) dac[u][j] = d4{0, 0, 0, 0};
    const d4 dz = {0, 0, 0, 0};
    float cv[12];
    h8 ha = {(_Float16)(1.f + s), (_Float16)0.5f, (_Float16)s, (_Float16)2.f, (_Float16)s, (_Float16)s, (_Float16)1.f, (_Float16)s}, hb = ha;
    h4 ha4 = {(_Float16)(1.f + s), (_Float16)s, (_Float16)s, (_Float16)s}, hb4 = ha4;
    f4 hacc[2][3];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 3; j++) hacc[u][j] = f4{0, 0, 0, 0};
    const f4 fz = {0, 0, 0, 0};
    f2 k10 = {9.765625e-4f + s, 9.765625e-4f + s};
// split-f16 exact form: per row one K = 32 fp16 MFMA for the cross terms, one (K = 32 or the legacy K = 16) for the exactly representable high parts, then
// E = D_cross 2^-10 + D_hi as a packed fma over the register pairs
#define HX(u, j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(hacc[u][j]) : "v"(ha), "v"(hb), "v"(fz))
#define HH(u, j) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %3" : "=v"(hacc[u][j]) : "v"(ha4), "v"(hb4), "v"(fz))
#define CMB(j, r) { f2 lo_ = {hacc[0][j][2 * (r)], hacc[0][j][2 * (r) + 1]}, hi_ = {hacc[1][j][2 * (r)], hacc[1][j][2 * (r) + 1]}; \
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi_) : "v"(lo_), "v"(k10)); q[(j) * 4 + 2 * (r)] += hi_.x * 1e-30f; }
#define MF(j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(bs), "v"(cs))
#define MD(u, j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %3" : "=v"(dac[u][j]) : "v"(bd), "v"(cd), "v"(dz))
#define CV(u, j, r) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(cv[(j) * 4 + (r)]) : "v"(dac[u][j][r]))
#define CB() asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(cd) : "v"(cs))
#define TR(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]))
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c))
#define PL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(bs))
#define TAIL()                                     \
    _Pragma("unroll") for (int i = 0; i < 16; i++) TR(i); \
    _Pragma("unroll") for (int i = 0; i < 15; i++) PK(i); \
    _Pragma("unroll") for (int i = 0; i < 14; i++) PL(i);
#define CVALL(u) _Pragma("unroll") for (int j = 0; j < 3; j++) { CV(u, j, 0); CV(u, j, 1); CV(u, j, 2); CV(u, j, 3); }
    if (FORM == 2) { CB(); MD(0, 0); MD(0, 1); MD(0, 2); }
    for (int it = 0; it < iters; it++) {
        if (FORM == 0) { MF(0); MF(1); MF(2); TAIL(); }
        if (FORM == 1) { CB(); MD(0, 0); MD(0, 1); MD(0, 2); CVALL(0); for (int i = 0; i < 12; i++) q[i] += cv[i] * 1e-30f; TAIL(); }
        if (FORM == 2) {
            // two m per iteration, ping-pong accumulators: issue m + 1's MFMAs, then convert + finish m
            CB(); MD(1, 0); MD(1, 1); MD(1, 2); CVALL(0); for (int i = 0; i < 12; i++) q[i] += cv[i] * 1e-30f; TAIL();
            CB(); MD(0, 0); MD(0, 1); MD(0, 2); CVALL(1); for (int i = 0; i < 12; i++) q[i] += cv[i] * 1e-30f; TAIL();
        }
        if (FORM == 3) { TAIL(); }                                             // the VALU tail alone
        if (FORM == 4) { MD(0, 0); MD(0, 1); MD(0, 2); }                        // three independent fp64 MFMAs alone
        if (FORM == 5) { CVALL(0); }                                           // twelve conversions alone
        if (FORM == 6) { HX(0, 0); HX(0, 1); HX(0, 2); HX(1, 0); HX(1, 1); HX(1, 2); CMB(0, 0); CMB(0, 1); CMB(1, 0); CMB(1, 1); CMB(2, 0); CMB(2, 1); TAIL(); }
        if (FORM == 7) { HX(0, 0); HX(0, 1); HX(0, 2); HH(1, 0); HH(1, 1); HH(1, 2); CMB(0, 0); CMB(0, 1); CMB(1, 0); CMB(1, 1); CMB(2, 0); CMB(2, 1); TAIL(); }
        if (FORM == 8) { HX(0, 0); HX(0, 1); HX(0, 2); HX(1, 0); HX(1, 1); HX(1, 2); }   // six K = 32 fp16 MFMAs alone
        if (FORM == 9) { HX(0, 0); HX(0, 1); HX(0, 2); HH(1, 0); HH(1, 1); HH(1, 2); }   // three K = 32 + three K = 16 alone
        if (FORM == 10) {  // the split form with the MFMAs of the next m issued before this m's tail (two m per iteration, their accumulators renamed by the asm outputs)
            HX(0, 0); HX(0, 1); HX(0, 2); HH(1, 0); HH(1, 1); HH(1, 2); TAIL(); CMB(0, 0); CMB(0, 1); CMB(1, 0); CMB(1, 1); CMB(2, 0); CMB(2, 1);
        }
    }
    float sum = 0;
    for (int j = 0; j < 3; j++) sum += acc[j].x + acc[j].y + acc[j].z + acc[j].w + (float)(dac[0][j].x + dac[0][j].w + dac[1][j].y);
#pragma unroll
    for (int i = 0; i < 16; i++) sum += a[i].x + a[i].y + q[i];
    for (int i = 0; i < 12; i++) sum += cv[i];
    for (int u = 0; u < 2; u++) for (int j = 0; j < 3; j++) sum += hacc[u][j].x + hacc[u][j].y + hacc[u][j].z + hacc[u][j].w;
    if (sum == 1234.5678f) out[threadIdx.x] = sum + (float)cd;
}

template <int FORM>
static void run_mix(int wg_per_cu, int iters, float* out) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix<FORM>, dim3(grid), dim3(256), 0, 0, out, iters / 10, 0.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mix<FORM>, dim3(grid), dim3(256), 0, 0, out, iters, 0.f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static const char* nm[] = {"fast: 3 MFMA f32 + tail", "exact: cvt B + 3 MFMA f64 + 12 cvt + tail", "exact, next m's MFMAs issued before this m's tail",
                               "tail alone (16 rsq + 15 pk_fma + 14 mul)", "3 MFMA f64 alone", "12 v_cvt_f32_f64 alone",
                               "split f16: 6 MFMA 16x16x32_f16 + 6 pk_fma + tail", "split f16: 3 MFMA x32 + 3 MFMA x16 + 6 pk_fma + tail", "6 MFMA 16x16x32_f16 alone",
                               "3 MFMA 16x16x32_f16 + 3 MFMA 16x16x16_f16 alone", "split f16 (x32 + x16), tail between MFMAs and combine"};
    const double per = (FORM == 2) ? 2.0 : 1.0;
    printf("K2 mix per 16 hyp x 16 px, %-58s waves/SIMD %d: %7.1f cycles per SIMD at 2.4 GHz by wall\n", nm[FORM], wg_per_cu,
           ms * 1e-3 * 2.4e9 / ((double)iters * per * wg_per_cu));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    CK(hipMalloc(&out, 4096));
    layout();
    numerics();
    if (argc > 2) return 0;
    for (int w : {1, 2, 4}) run<R_MFMA64>(w, iters, out);
    for (int w : {1, 2}) run<R_MFMA64_DEP>(w, iters, out);
    for (int w : {1, 4}) run<R_MFMA32>(w, iters, out);
    for (int w : {1, 2, 4}) run<R_CVT_F32_F64>(w, iters, out);
    for (int w : {1, 4}) run<R_CVT_F64_F32>(w, iters, out);
    for (int w : {1, 4}) run<R_FMA64>(w, iters, out);
    for (int w : {1, 2, 3, 4}) {
        run_mix<0>(w, iters, out); run_mix<1>(w, iters, out); run_mix<2>(w, iters, out);
        run_mix<3>(w, iters, out); run_mix<4>(w, iters, out); run_mix<5>(w, iters, out);
        run_mix<6>(w, iters, out); run_mix<7>(w, iters, out); run_mix<8>(w, iters, out); run_mix<9>(w, iters, out); run_mix<10>(w, iters, out);
    }
    return 0;
}

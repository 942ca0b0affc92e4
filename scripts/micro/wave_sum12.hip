#include <hip/hip_runtime.h>
__device__ __forceinline__ void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_merge(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, BANK, false));
}
// 12 values per lane -> sums over the 64 lanes; lane L ends with the total of value index
//   6*(L>>5) + 3*((L>>4)&1) + {0: L&3==0, 1: L&3==2, 2: L&3==1}   (L&3 == 3 holds nothing)
__device__ __forceinline__ float wave_sum12(float (&a)[12]) {
#pragma unroll
    for (int k = 0; k < 6; k++) { swap32(a[k], a[6 + k]); a[k] += a[6 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { swap16(a[k], a[3 + k]); a[k] += a[3 + k]; }
    const bool odd = threadIdx.x & 1, b1 = threadIdx.x & 2;
    // lane ^ 1: even lanes keep (a0, a1), odd lanes keep (a2, -); each lane sends what its partner keeps
    const float s0 = (odd ? a[2] : a[0]) + dpp_merge<0xB1, 0xf>(0.f, odd ? a[0] : a[2]);
    const float s1 = (odd ? 0.f : a[1]) + dpp_merge<0xB1, 0xf>(0.f, odd ? a[1] : 0.f);
    // lane ^ 2: bit1 = 0 keeps s0, bit1 = 1 keeps s1
    float v = (b1 ? s1 : s0) + dpp_merge<0x4E, 0xf>(0.f, b1 ? s0 : s1);
    v += dpp_merge<0x124, 0xf>(0.f, v);  // row_ror:4
    v += dpp_merge<0x128, 0xf>(0.f, v);  // row_ror:8
    return v;
}
__global__ void k(float* o, const float* in) {
    float a[12];
    for (int i = 0; i < 12; i++) a[i] = in[i * 64 + threadIdx.x];
    o[threadIdx.x] = wave_sum12(a);
}
#include <cstdio>
#include <cmath>
#include <cstring>
int main() {
    static float h[12 * 64], out[64];
    static int cnt[64][12];          // how many lanes of value i reach output lane L
    static unsigned long long msk[64][12];
    float *din, *dout;
    (void)hipMalloc(&din, sizeof h); (void)hipMalloc(&dout, sizeof out);
    for (int i = 0; i < 12; i++) for (int l = 0; l < 64; l++) {
        memset(h, 0, sizeof h); h[i * 64 + l] = 1.f;
        (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dout, din);
        (void)hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
        for (int L = 0; L < 64; L++) if (out[L] != 0.f) { cnt[L][i] += (int)out[L]; msk[L][i] |= 1ull << l; }
    }
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        const int m = L & 3, expect = (m == 3) ? -1 : 6 * (L >> 5) + 3 * ((L >> 4) & 1) + (m == 0 ? 0 : m == 2 ? 1 : 2);
        printf("lane %2d (expect %2d):", L, expect);
        int ok = expect < 0;
        for (int i = 0; i < 12; i++) if (cnt[L][i]) { printf("  v%d x%d [%016llx]", i, cnt[L][i], msk[L][i]); if (i == expect && cnt[L][i] == 64 && msk[L][i] == ~0ull) ok = 1; else if (expect >= 0) ok = -100; }
        if (ok <= 0) bad++;
        printf("%s\n", ok > 0 ? "" : "   MISMATCH");
    }
    printf("wave_sum12 mapping: %s\n", bad ? "DIFFERENT" : "as documented");
    return bad != 0;
}

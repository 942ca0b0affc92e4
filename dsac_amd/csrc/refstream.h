// refstream.h -- the reference's random stream, restated (round 6).
//
// The reference draws its minimal sets from ThreadRand (core/thread_rand.cpp:40-69): one std::mt19937 per OpenMP thread, generator t seeded with
// seed + t, and irand(lo, hi) = std::uniform_int_distribution<int>(lo, hi - 1) on it (:59-69, :95-98).  The sampling loop (core/cnn_softam.h:1010-1060)
// draws x = irand(0, cols) BEFORE y = irand(0, rows), re-draws a cell that is already in the set, and starts a new attempt after a failed P3P or
// re-projection check.  Everything in this header is a pure function of the 32-bit output stream of the generator, so it runs the same on the host
// and in a kernel; the kernels are in k_refstream.hip.
//
//   mt_seed / mt_twist_word / mt_temper : MT19937 (Matsumoto & Nishimura 1998) as std::mt19937 parameterises it (w 32, n 624, m 397, r 31, a 0x9908b0df,
//                                         u 11, s 7, b 0x9d2c5680, t 15, c 0xefc60000, l 18, f 1812433253)
//   bounded(next, n, mode)              : one draw of std::uniform_int_distribution<int>(0, n - 1) from a 32-bit generator, as libstdc++ computes it:
//                                         mode 0 = GCC >= 11 (bits/uniform_int_dist.h _S_nd: Lemire's nearly divisionless method on the 64-bit product),
//                                         mode 1 = GCC <= 10 (scaling = (2^32 - 1) / n, reject >= n * scaling, divide)
//   parse_attempt(raw, ...)             : one attempt's four distinct cells from the raw stream; returns the number of outputs it consumed
//
// Pinned: tests/test_refstream.py compiles this header with g++ and compares bounded() with the standard library's own distribution over both modes'
// edge cases, and the attempt parser with the oracle's loop (which uses the standard library itself) -- which in turn reproduces the REAL reference's
// minimal sets on both golden frames (tests/test_reference_pinning.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RS_FN __host__ __device__ __forceinline__
#else
#define RS_FN inline
#endif

namespace rs {

constexpr int MT_N = 624, MT_M = 397;

RS_FN uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// x[k + n] from x[k], x[k + 1], x[k + m]
RS_FN uint32_t mt_twist_word(uint32_t cur, uint32_t next, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (next & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
RS_FN void mt_seed(uint32_t* mt, uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}
// the whole next block in place, one word after the other (the host-side / single-lane form; the kernels twist in three parallel phases)
RS_FN void mt_twist_block(uint32_t* mt) {
    for (int k = 0; k < MT_N - MT_M; k++) mt[k] = mt_twist_word(mt[k], mt[k + 1], mt[k + MT_M]);
    for (int k = MT_N - MT_M; k < MT_N - 1; k++) mt[k] = mt_twist_word(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
    mt[MT_N - 1] = mt_twist_word(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
}

// std::uniform_int_distribution<int>(0, n - 1)(g) for a generator with range [0, 2^32 - 1] (std::mt19937: result_type is 64 bits wide on LP64, its range is
// 32 bits, which is what selects this branch in libstdc++).  next() yields the next raw output.
template <class Next>
RS_FN uint32_t bounded(Next&& next, uint32_t n, int mode) {
    if (mode == 0) {
        uint64_t product = (uint64_t)next() * (uint64_t)n;
        uint32_t low = (uint32_t)product;
        if (low < n) {
            const uint32_t threshold = (0u - n) % n;
            while (low < threshold) {
                product = (uint64_t)next() * (uint64_t)n;
                low = (uint32_t)product;
            }
        }
        return (uint32_t)(product >> 32);
    }
    const uint32_t scaling = 0xffffffffu / n, past = n * scaling;
    uint32_t r;
    do r = next(); while (r >= past);
    return r / scaling;
}

// One attempt of the sampling loop: cells until four distinct ones, x before y.  raw(i) = raw output i of the stream (i < avail).
// Returns the outputs consumed (>= 8), -1 when the stream ran out before the attempt was complete, -2 when 64 candidate cells did not give four distinct
// ones (a map with fewer than four cells: the reference would loop forever).
template <class Raw>
RS_FN int parse_attempt(Raw&& raw, long long pos, long long avail, uint32_t W, uint32_t H, int mode, int32_t set4[4]) {
    long long p = pos;
    bool starved = false;
    auto next = [&]() -> uint32_t {
        // out of stream: a value both rejection loops accept at once (Lemire: low = 2^32 - n >= n; scaling: 0 < past), so that they end
        if (p >= avail) { starved = true; p++; return mode == 0 ? 0xffffffffu : 0u; }
        return raw(p++);
    };
    int cnt = 0;
    set4[0] = set4[1] = set4[2] = set4[3] = -1;
    for (int cand = 0; cnt < 4; cand++) {
        if (cand >= 64) return -2;
        const uint32_t x = bounded(next, W, mode);
        const uint32_t y = bounded(next, H, mode);
        if (starved) return -1;
        const int32_t idx = (int32_t)(y * W + x);
        if (idx == set4[0] || idx == set4[1] || idx == set4[2]) continue;  // cnt < 4: set4[3] is still free
        if (cnt == 0) set4[0] = idx; else if (cnt == 1) set4[1] = idx; else if (cnt == 2) set4[2] = idx; else set4[3] = idx;
        cnt++;
    }
    return (int)(p - pos);
}

// #pragma omp parallel for, default (static) schedule, N iterations on T threads (libgomp): thread t runs [t q + min(t, r), + q + (t < r)), q = N / T, r = N % T
RS_FN void static_chunk(int N, int T, int t, int& first, int& count) {
    const int q = N / T, r = N % T;
    first = t * q + (t < r ? t : r);
    count = q + (t < r ? 1 : 0);
}

}  // namespace rs

// Test harness (tests/test_refstream.py): dsac_amd/csrc/refstream.h compiled for the host, next to the standard library's own generator and distribution.
#include "../../dsac_amd/csrc/refstream.h"
#include <cstdint>
#include <random>
#include <vector>

extern "C" {

// n raw outputs of the restated generator after `skip`
void rsh_raw(uint32_t seed, uint64_t skip, int n, uint32_t* out) {
    std::vector<uint32_t> mt(rs::MT_N);
    rs::mt_seed(mt.data(), seed);
    int idx = rs::MT_N;
    for (uint64_t i = 0; i < skip + (uint64_t)n; i++) {
        if (idx == rs::MT_N) { rs::mt_twist_block(mt.data()); idx = 0; }
        const uint32_t v = rs::mt_temper(mt[idx++]);
        if (i >= skip) out[i - skip] = v;
    }
}
void rsh_raw_std(uint32_t seed, uint64_t skip, int n, uint32_t* out) {
    std::mt19937 g(seed);
    g.discard(skip);
    for (int i = 0; i < n; i++) out[i] = (uint32_t)g();
}

// count draws of bounded(n) from a given raw stream: restated (mode) and the standard library's distribution on a generator replaying the same raw values
struct Replay {
    typedef std::mt19937::result_type result_type;
    const uint32_t* raw; long long pos = 0;
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return 0xffffffffu; }
    result_type operator()() { return raw[pos++]; }
};
void rsh_bounded(const uint32_t* raw, int count, uint32_t n, int mode, uint32_t* out, long long* used) {
    long long p = 0;
    auto next = [&]() { return raw[p++]; };
    for (int i = 0; i < count; i++) out[i] = rs::bounded(next, n, mode);
    *used = p;
}
void rsh_bounded_std(const uint32_t* raw, int count, uint32_t n, uint32_t* out, long long* used) {
    Replay g{raw};
    for (int i = 0; i < count; i++) out[i] = (uint32_t)std::uniform_int_distribution<int>(0, (int)n - 1)(g);
    *used = g.pos;
}
int rsh_glibcxx_release() {
#ifdef _GLIBCXX_RELEASE
    return _GLIBCXX_RELEASE;
#else
    return 0;
#endif
}

// attempts parsed one after the other from the restated stream
int rsh_attempts(uint32_t seed, uint64_t skip, int A, int W, int H, int mode, int32_t* sets, long long* offs) {
    const int D = 12 * A + 64;
    std::vector<uint32_t> raw(D);
    rsh_raw(seed, skip, D, raw.data());
    long long cur = 0;
    for (int a = 0; a < A; a++) {
        const int len = rs::parse_attempt([&](long long i) { return raw[i]; }, cur, (long long)D, (uint32_t)W, (uint32_t)H, mode, sets + 4 * a);
        if (len <= 0) return a;
        offs[a] = cur;
        cur += len;
    }
    offs[A] = cur;
    return A;
}
void rsh_static_chunk(int N, int T, int t, int* first, int* count) { rs::static_chunk(N, T, t, *first, *count); }
}

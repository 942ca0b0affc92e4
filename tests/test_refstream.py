"""The reference's random stream, restated (dsac_amd/csrc/refstream.h; round 6): MT19937, std::uniform_int_distribution<int> as libstdc++ computes it, the
attempt parser of the sampling loop (core/cnn_softam.h:1010-1060, core/thread_rand.cpp:40-69).  CPU tests: the header is compiled with g++ next to the
standard library's own generator and distribution; the oracle's loop (which uses the standard library itself) reproduces the REAL reference's minimal
sets on both golden frames (the fixtures of tests/golden/make_golden_ref.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def rsh(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("rsh") / "librsh.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", out, os.path.join(HERE, "helpers", "refstream_host.cpp")])
    return C.CDLL(out)


def u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def test_generator_equals_std_mt19937(rsh):
    for seed, skip in ((1305, 0), (1305, 6400), (4242, 623), (0, 624), (0xFFFFFFFF, 1247), (5489, 100000)):
        a, b = np.zeros(3000, np.uint32), np.zeros(3000, np.uint32)
        rsh.rsh_raw(C.c_uint32(seed), C.c_uint64(skip), 3000, u32p(a))
        rsh.rsh_raw_std(C.c_uint32(seed), C.c_uint64(skip), 3000, u32p(b))
        assert np.array_equal(a, b), (seed, skip)
    # the 10000th output of mt19937() is 4123659995 (ISO C++ [rand.predef])
    a = np.zeros(1, np.uint32)
    rsh.rsh_raw(C.c_uint32(5489), C.c_uint64(9999), 1, u32p(a))
    assert int(a[0]) == 4123659995


def test_bounded_equals_std_uniform_int_distribution(rsh):
    """The mode this toolchain's libstdc++ uses must match std::uniform_int_distribution draw for draw, rejections included: the raw stream is seeded with
    the values that trigger them (low products for Lemire's method, values past n * scaling for the division method)."""
    rel = rsh.rsh_glibcxx_release()
    mode = 0 if rel >= 11 else 1
    rng = np.random.default_rng(7)
    for n in (40, 640, 480, 3, 7, 1000, 65537, 2 ** 30 + 3, 2 ** 31 - 1):
        raw = rng.integers(0, 2 ** 32, 6000, dtype=np.uint64).astype(np.uint32)
        # adversarial values: products whose low word is tiny (Lemire rejects below (2^32 - n) % n), and the top of the range (the division method rejects there)
        k = np.arange(1, 400, dtype=np.uint64)
        raw[10:10 + 399] = ((k * (2 ** 32) + n - 1) // n).astype(np.uint32)  # smallest r with floor(r n / 2^32) = k: low word < n
        raw[1000:1200] = (2 ** 32 - 1 - np.arange(200)).astype(np.uint32)
        raw[2000:2050] = np.arange(50, dtype=np.uint32)
        a, b = np.zeros(4000, np.uint32), np.zeros(4000, np.uint32)
        ua, ub = C.c_longlong(0), C.c_longlong(0)
        rsh.rsh_bounded(u32p(raw), 4000, C.c_uint32(n), mode, u32p(a), C.byref(ua))
        rsh.rsh_bounded_std(u32p(raw), 4000, C.c_uint32(n), u32p(b), C.byref(ub))
        assert np.array_equal(a, b) and ua.value == ub.value, "n = %d, libstdc++ %d" % (n, rel)
        assert a.max() < n
    # the other mode is a different function of the stream (so the option matters): visibly so for a large range, almost never for a map's width
    raw = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    ua = C.c_longlong(0)
    for n, same in ((2 ** 30 + 3, False), (640, True)):
        a, b = np.zeros(3000, np.uint32), np.zeros(3000, np.uint32)
        rsh.rsh_bounded(u32p(raw), 3000, C.c_uint32(n), 0, u32p(a), C.byref(ua))
        rsh.rsh_bounded(u32p(raw), 3000, C.c_uint32(n), 1, u32p(b), C.byref(ua))
        assert a.max() < n and b.max() < n
        assert ((a == b).mean() > 0.99) == same


def test_division_mode_by_its_definition(rsh):
    """mode 1 (libstdc++ <= 10): scaling = (2^32 - 1) / n, reject r >= n * scaling, r / scaling."""
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    raw[:64] = (2 ** 32 - 1 - np.arange(64)).astype(np.uint32)
    for n in (40, 640, 480, 1000):
        a = np.zeros(4000, np.uint32)
        ua = C.c_longlong(0)
        rsh.rsh_bounded(u32p(raw), 4000, C.c_uint32(n), 1, u32p(a), C.byref(ua))
        scaling = (2 ** 32 - 1) // n
        want, p = [], 0
        while len(want) < 4000:
            r = int(raw[p]); p += 1
            if r >= n * scaling:
                continue
            want.append(r // scaling)
        assert np.array_equal(a, np.array(want, np.uint32)) and ua.value == p


def test_static_schedule(rsh):
    for N, T in ((64, 1), (64, 4), (64, 3), (64, 7), (256, 16), (5, 8)):
        cover = []
        for t in range(T):
            f, c = C.c_int(0), C.c_int(0)
            rsh.rsh_static_chunk(N, T, t, C.byref(f), C.byref(c))
            assert c.value in (N // T, N // T + 1)
            cover += list(range(f.value, f.value + c.value))
        assert cover == list(range(N))


def test_oracle_loop_reproduces_the_real_reference_sets(orc):
    """orc_sample_refstream (std::mt19937(seed) + std::uniform_int_distribution, single thread) with the 6400 outputs of stochasticSubSample skipped
    (two drand = four outputs per cell of its 40 x 40 grid, core/cnn_softam.h:283-309) draws the minimal sets the real processImage drew, accepts the ones it
    accepted, and computes the same P3P poses."""
    for v, seed in ((1, 1305), (2, 4242)):
        g = np.load(os.path.join(HERE, "golden", "ref_frame_v%d.npz" % v))
        sets_ref = g["sampledPoints"][:, :, 1] * 40 + g["sampledPoints"][:, :, 0]
        p, s, ok, cons, att = orc.sample_refstream(64, seed, g["estObj"].astype(np.float32), g["sampling"].astype(np.float32), 40, 40, g["cam"], threads=1, skip32=[6400])
        assert ok.all() and np.array_equal(s, sets_ref)
        assert np.abs(p - g["hyps"]).max() <= 1e-9
        assert att[0] >= 64 and cons[0] >= 8 * att[0]
        # without the skip the stream is another one
        _, s0, _, _, _ = orc.sample_refstream(64, seed, g["estObj"].astype(np.float32), g["sampling"].astype(np.float32), 40, 40, g["cam"], threads=1)
        assert not np.array_equal(s0, sets_ref)


def test_attempt_parser_equals_the_oracle_loop(rsh, orc):
    """The restated parser walks the same attempts as the oracle's loop: every set the oracle ACCEPTS appears, in order, among the parsed attempts, and the
    outputs consumed up to the last accepted attempt agree (40 x 40: duplicates cells occur, so attempts of more than 8 outputs are exercised)."""
    rel = rsh.rsh_glibcxx_release()
    mode = 0 if rel >= 11 else 1
    g = np.load(os.path.join(HERE, "golden", "ref_frame_v2.npz"))
    xyz, uv = g["estObj"].astype(np.float32), g["sampling"].astype(np.float32)
    p, s, ok, cons, att = orc.sample_refstream(64, 4242, xyz, uv, 40, 40, g["cam"], threads=1, skip32=[6400])
    A = int(att[0])
    sets = np.zeros((A, 4), np.int32)
    offs = np.zeros(A + 1, np.int64)
    n = rsh.rsh_attempts(C.c_uint32(4242), C.c_uint64(6400), A, 40, 40, mode, sets.ctypes.data_as(C.POINTER(C.c_int32)), offs.ctypes.data_as(C.POINTER(C.c_longlong)))
    assert n == A and offs[A] == int(cons[0])
    assert (np.diff(offs) >= 8).all() and (np.diff(offs) > 8).any()  # some attempt re-drew a duplicate cell
    j = 0
    for a in range(A):
        if j < 64 and np.array_equal(sets[a], s[j]):
            j += 1
    assert j == 64
    assert np.array_equal(sets[A - 1], s[63])  # the loop stopped on its last accepted attempt


def test_oracle_loop_reproduces_the_real_reference_with_several_openmp_threads(orc):
    """The REAL reference run with omp_set_num_threads(T), T = 3 and 4 (tests/golden/make_golden_ref.py -> ref_threads_v<K>.npz: its own `#pragma omp
    parallel for` with the static schedule, thread t drawing from mt19937(seed + t), thread 0 behind the sub-sampler's 6400 outputs): the oracle's loop
    draws the same 64 sets and poses."""
    for v in (1, 2):
        g = np.load(os.path.join(HERE, "golden", "ref_frame_v%d.npz" % v))
        t = np.load(os.path.join(HERE, "golden", "ref_threads_v%d.npz" % v))
        for T in (3, 4):
            skip = np.zeros(T, np.uint64)
            skip[0] = 6400
            p, s, ok, cons, att = orc.sample_refstream(64, int(t["seed"]), g["estObj"].astype(np.float32), g["sampling"].astype(np.float32), 40, 40, g["cam"], threads=T, skip32=skip)
            assert ok.all() and np.array_equal(s, t["t%d_sets" % T]), (v, T)
            assert np.abs(p - t["t%d_hyps" % T]).max() <= 1e-9
            assert not np.array_equal(t["t%d_sets" % T], g["sampledPoints"][:, :, 1] * 40 + g["sampledPoints"][:, :, 0])  # another thread count, other sets

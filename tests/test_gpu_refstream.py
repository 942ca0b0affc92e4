"""dsac_sample_refstream (round 6): K1 drawing from the reference's own generators -- std::mt19937(seed + t) per OpenMP thread through
std::uniform_int_distribution (core/thread_rand.cpp:40-69), x before y, re-draw of a duplicate cell, a new attempt after a failed P3P / re-projection
check (core/cnn_softam.h:1010-1060).  The minimal sets are bit-identical to the REAL reference's processImage on both golden frames without passing
`sets` (one thread, as oracle/_ref runs it), and to the oracle's loop -- the standard library's generator and distribution themselves -- for several
thread counts, across successive calls on the running generators, and on a 640 x 480 map."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SUBSAMPLE_OUTPUTS = 6400  # stochasticSubSample (core/cnn_softam.h:283-309): two drand = four 32-bit outputs per cell of its 40 x 40 grid, on thread 0


def golden(v):
    g = np.load(os.path.join(HERE, "golden", "ref_frame_v%d.npz" % v))
    sets_ref = g["sampledPoints"][:, :, 1] * 40 + g["sampledPoints"][:, :, 0]
    return g, sets_ref, g["estObj"].astype(np.float32), g["sampling"].astype(np.float32)


@pytest.mark.parametrize("v,seed", [(1, 1305), (2, 4242)])
def test_sets_identical_to_the_real_reference_without_replay(engine, v, seed):
    g, sets_ref, xyz, uv = golden(v)
    engine.set_frame(xyz, uv, 40, 40, g["cam"])
    engine.refstreamInit(seed, 1)
    engine.refstreamDiscard(0, SUBSAMPLE_OUTPUTS)
    poses, sets, ok, consumed, attempts = engine.sampleRefstream(64, thr=10.0)
    assert ok.all()
    assert np.array_equal(sets, sets_ref), "minimal sets differ from the real reference's (%d of 64 equal)" % (sets == sets_ref).all(axis=1).sum()
    assert np.abs(poses - g["hyps"]).max() <= 1e-5  # P3P poses of the real reference (cv convention)
    assert attempts[0] >= 64 and consumed[0] >= 8 * attempts[0]
    # the same sets through the replay path give the same poses bit for bit (one evaluation kernel behind both)
    p2, s2, ok2 = engine.sample(64, sets=sets_ref, thr=10.0)
    assert np.array_equal(p2, poses) and ok2.all()


@pytest.mark.parametrize("v", [1, 2])
@pytest.mark.parametrize("threads", [3, 4])
def test_sets_identical_to_the_real_reference_run_with_several_openmp_threads(engine, v, threads):
    """tests/golden/ref_threads_v<K>.npz: the REAL reference's processImage run with omp_set_num_threads(T) -- its sets for T = 3 and 4, reproduced on the
    device from nothing but the seed and the thread count."""
    g, _, xyz, uv = golden(v)
    t = np.load(os.path.join(HERE, "golden", "ref_threads_v%d.npz" % v))
    engine.set_frame(xyz, uv, 40, 40, g["cam"])
    engine.refstreamInit(int(t["seed"]), threads)
    engine.refstreamDiscard(0, SUBSAMPLE_OUTPUTS)
    poses, sets, ok, consumed, attempts = engine.sampleRefstream(64, thr=10.0)
    assert ok.all() and np.array_equal(sets, t["t%d_sets" % threads])
    assert np.abs(poses - t["t%d_hyps" % threads]).max() <= 1e-5


@pytest.mark.parametrize("threads", [1, 3, 4, 7])
def test_threads_and_running_generators_against_the_oracle(engine, orc, threads):
    g, _, xyz, uv = golden(2)
    engine.set_frame(xyz, uv, 40, 40, g["cam"])
    skip = np.zeros(threads, np.uint64)
    skip[0] = SUBSAMPLE_OUTPUTS
    engine.refstreamInit(4242, threads)
    engine.refstreamDiscard(0, SUBSAMPLE_OUTPUTS)
    for call in range(3):  # three images from the running generators, as the reference's loop over its test images
        po, so, oko, co, ao = orc.sample_refstream(61, 4242, xyz, uv, 40, 40, g["cam"], threads=threads, skip32=skip)
        p, s, ok, c, a = engine.sampleRefstream(61, thr=10.0)
        assert np.array_equal(s, so) and np.array_equal(ok, oko), "call %d" % call
        assert np.array_equal(c, co) and np.array_equal(a, ao)
        assert np.abs(p - po).max() <= 1e-6
        skip = skip + co


def test_640x480_and_small_windows(engine, orc, synth):
    H, W = 480, 640
    fr = synth.chess_like_frame(H, W, seed=2305)
    uv = synth.pixel_grid(H, W)
    engine.set_frame(fr["xyz"], None, H, W, fr["cam"])
    po, so, oko, co, ao = orc.sample_refstream(256, 1305, fr["xyz"], uv, H, W, fr["cam"], threads=1)
    engine.refstreamInit(1305, 1)
    p, s, ok, c, a = engine.sampleRefstream(256, thr=10.0)
    assert np.array_equal(s, so) and ok.all() and np.array_equal(c, co) and np.array_equal(a, ao)
    # a budget smaller than what the stream needs: the served hypotheses are the same ones, the rest report ok = 0, the generator stops where the budget ends
    engine.refstreamInit(1305, 1)
    budget = int(ao[0]) // 2
    p2, s2, ok2, c2, a2 = engine.sampleRefstream(256, thr=10.0, max_attempts=budget)
    n = int(ok2.sum())
    assert 0 < n < 256 and ok2[:n].all() and not ok2[n:].any()
    assert np.array_equal(s2[:n], so[:n]) and a2[0] == budget


def test_misuse(engine):
    import dsac_amd
    from dsac_amd.capi import lib, ptr
    g, _, xyz, uv = golden(1)
    e2 = dsac_amd.Engine(0)
    try:
        e2.set_frame(xyz, uv, 40, 40, g["cam"])
        with pytest.raises(dsac_amd.capi.DsacError):
            e2.sampleRefstream(8)  # no dsac_refstream_init
        e2.refstreamInit(1, 2)
        with pytest.raises(dsac_amd.capi.DsacError):
            e2.refstreamDiscard(2, 10)
        with pytest.raises(dsac_amd.capi.DsacError):
            e2.sampleRefstream(8, max_attempts=0)
        with pytest.raises(dsac_amd.capi.DsacError):
            e2.set_option("refstream_mode", 5)
        # the other libstdc++'s distribution is another function of the stream only where it rejects or at bucket edges: on a 40 x 40 map the sets agree
        e2.set_option("refstream_mode", 1)
        e2.refstreamInit(1305, 1)
        _, s1, _, _, _ = e2.sampleRefstream(16)
        e2.set_option("refstream_mode", 0)
        e2.refstreamInit(1305, 1)
        _, s0, _, _, _ = e2.sampleRefstream(16)
        assert s0.shape == s1.shape
    finally:
        e2.close()

"""Edge cases of the C ABI on the GPU: empty and degenerate inputs, NaNs, big and re-sized frames, several contexts."""
import numpy as np
import pytest

import dsac_amd
from conftest import excl_clamp_edge

pytestmark = pytest.mark.gpu


def test_empty_batches_are_no_ops(engine, frame40):
    fr = frame40
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    p, s, ok = engine.sample(0)
    assert p.shape == (0, 6)
    engine.reproject(np.zeros((0, 6)), N=0, err=np.zeros((0, 1600), np.float32))
    assert engine.dPNP(np.zeros((0, 4), np.int32)).shape == (0, 6, 12)
    g = engine.dScore(np.zeros((0, 6)), np.zeros((0, 4), np.int32), np.zeros((0, 1600), np.float32))
    assert np.all(g == 0)
    out, sd = engine.refine(np.zeros((0, 6)), np.zeros((1, 1600), np.int32))
    assert out.shape == (0, 6)
    with pytest.raises(dsac_amd.capi.DsacError):
        engine.softMax(np.zeros(0))  # softmax of nothing is an error, like an empty std::vector in the reference would be UB


def test_tiny_maps(engine, orc):
    # fewer than 4 cells: sampling is impossible, reprojection still works
    xyz = np.array([[0, 0, 1000], [10, 0, 1000], [0, 10, 1000]], np.float32)
    uv = np.array([[320, 240], [330, 240], [320, 250]], np.float32)
    engine.set_frame(xyz, uv, 1, 3, (525.0, 525.0, 320.0, 240.0))
    with pytest.raises(dsac_amd.capi.DsacError):
        engine.sample(4)
    e = engine.getDiffMap(np.zeros((1, 6))).reshape(-1)
    ref = orc.get_diff_maps(np.zeros(6), xyz, uv, 1, 3, (525.0, 525.0, 320.0, 240.0))[0]
    assert np.abs(e - ref).max() <= 1e-3
    # exactly 4 cells: every attempt uses all of them
    fr4 = dict(xyz=np.array([[0, 0, 1000], [100, 0, 1100], [0, 100, 1200], [100, 100, 900]], np.float32),
               uv=np.array([[320, 240], [367.7, 240], [320, 283.75], [378.3, 298.3]], np.float32))
    engine.set_frame(fr4["xyz"], fr4["uv"], 2, 2, (525.0, 525.0, 320.0, 240.0))
    p, s, ok = engine.sample(8, seed=1, max_tries=64)
    pr, sr, okr, _ = orc.sample(8, 1, fr4["xyz"], fr4["uv"], 2, 2, (525.0, 525.0, 320.0, 240.0), max_tries=64)
    assert np.array_equal(ok, okr)
    assert all(sorted(x) == [0, 1, 2, 3] for x in s[ok.astype(bool)])


def test_nan_and_inf_coordinates_do_not_poison_neighbours(engine, orc, frame40):
    fr = dict(frame40)
    xyz = fr["xyz"].copy()
    xyz[5] = np.nan
    xyz[77, 2] = np.inf
    engine.set_frame(xyz, fr["uv"], 40, 40, fr["cam"])
    poses, *_ = orc.sample(16, 3, frame40["xyz"], frame40["uv"], 40, 40, fr["cam"])
    e = engine.getDiffMap(poses).reshape(16, -1)
    ref = orc.get_diff_maps(poses, frame40["xyz"], fr["uv"], 40, 40, fr["cam"])
    good = np.ones(1600, bool)
    good[[5, 77]] = False
    m = excl_clamp_edge(e[:, good], ref[:, good])
    assert np.abs(e[:, good] - ref[:, good])[m].max() <= 1e-3
    soft = engine.softInlierScores(poses)
    assert np.all(np.isfinite(soft))  # NaN residuals clamp to 100 px (v_min_f32) and contribute ~0
    # a NaN pose gives a finite (clamped) error image and never crashes
    bad = poses.copy()
    bad[0, 0] = np.nan
    eb = engine.getDiffMap(bad).reshape(16, -1)
    assert np.all(np.isfinite(eb[1:])) and np.array_equal(eb[1:], e[1:])


def test_many_hypotheses_and_big_frame(engine, orc, synth):
    fr = synth.chess_like_frame(40, 40, seed=5, quantise_int16=True)
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    N = 20000
    poses, sets, ok = engine.sample(N, seed=11, max_tries=256)
    assert ok.mean() > 0.99
    soft = engine.softInlierScores(poses)
    w, ent, avg = engine.softMax(soft, 0.1, poses)
    assert abs(w.sum() - 1) < 1e-9 and np.all(np.isfinite(avg))
    idx = np.random.default_rng(0).choice(N, 32, replace=False)
    ref = orc.soft_inlier(orc.get_diff_maps(poses[idx], fr["xyz"], fr["uv"], 40, 40, fr["cam"]), 10.0, 0.5)
    assert np.abs(soft[idx] - ref).max() <= 1e-4 * max(1.0, ref.max())
    # full-HD map (P = 2 073 600): a few hypotheses, checked on a random subset of cells
    big = synth.chess_like_frame(1080, 1920, seed=9, cam=(1400.0, 1400.0, 960.0, 540.0))
    uvb = np.stack(np.meshgrid(np.arange(1920, dtype=np.float32), np.arange(1080, dtype=np.float32)), -1).reshape(-1, 2)
    engine.set_frame(big["xyz"], None, 1080, 1920, big["cam"])
    p3 = np.stack([big["gt_pose"], big["gt_pose"] + [0.01, 0, 0, 5, 0, 0], np.zeros(6)])
    e = engine.getDiffMap(p3).reshape(3, -1)
    cells = np.random.default_rng(1).choice(1080 * 1920, 5000, replace=False)
    ref = orc.get_diff_maps(p3, big["xyz"][cells], uvb[cells], 1, 5000, big["cam"])
    m = excl_clamp_edge(e[:, cells], ref)
    assert np.abs(e[:, cells] - ref)[m].max() <= 2e-3  # larger focal length and coordinates: fp32 projection error grows with f


@pytest.mark.parametrize("variant", [53, 55, 57, 65, 69, 72])
def test_per_wave_sum_forms_on_a_map_whose_chunk_count_is_not_a_multiple_of_the_waves(engine, orc, synth, variant):
    """The per-wave partial-sum forms with several waves per workgroup write PT * WAVES rows of partial sums; on the reference's 40 x 40 map
    (25 chunks of 64 cells) that exceeds ceil(P / 64) -- the scratch must hold them (it held 25 rows for 28) and the idle waves' zero rows
    must not disturb the sums."""
    fr = synth.chess_like_frame(40, 40, seed=5, quantise_int16=True)
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    poses, sets, ok, _ = orc.sample(128, 3, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    ref_err = orc.get_diff_maps(poses, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    ref = orc.soft_inlier(ref_err, 10.0, 0.5)
    engine.set_option("k2_variant", variant)
    try:
        for rep in range(3):  # an overflow would corrupt a neighbouring allocation: repeat and compare everything
            err = np.zeros((128, 1600), np.float32)
            soft = np.zeros(128)
            engine.reproject(poses, err=err, soft=soft)
            assert np.abs(soft - ref).max() <= 1e-4 * max(1.0, ref.max())
            m = excl_clamp_edge(err, ref_err)
            assert np.abs(err - ref_err)[m].max() <= 1e-3
    finally:
        engine.set_option("k2_variant", -1)


def test_frames_can_be_replaced_and_contexts_are_independent(orc, synth):
    a = synth.chess_like_frame(40, 40, seed=1)
    b = synth.chess_like_frame(24, 56, seed=2)
    with dsac_amd.Engine(0) as e1, dsac_amd.Engine(0) as e2:
        e1.set_frame(a["xyz"], a["uv"], 40, 40, a["cam"])
        e2.set_frame(b["xyz"], b["uv"], 24, 56, b["cam"])
        pa, sa, oka = e1.sample(64, seed=3)
        pb, sb, okb = e2.sample(64, seed=3)
        assert np.array_equal(sa, orc.sample(64, 3, a["xyz"], a["uv"], 40, 40, a["cam"])[1])
        assert np.array_equal(sb, orc.sample(64, 3, b["xyz"], b["uv"], 24, 56, b["cam"])[1])
        # swap the frames: contexts regrow their scratch and carry no state from the previous frame
        e1.set_frame(b["xyz"], b["uv"], 24, 56, b["cam"])
        e2.set_frame(a["xyz"], a["uv"], 40, 40, a["cam"])
        assert np.array_equal(e1.sample(64, seed=3)[1], sb)
        assert np.array_equal(e2.sample(64, seed=3)[1], sa)
        ea = e2.getDiffMap(pa).reshape(64, -1)
        ref = orc.get_diff_maps(pa, a["xyz"], a["uv"], 40, 40, a["cam"])
        m = excl_clamp_edge(ea, ref)
        assert np.abs(ea - ref)[m].max() <= 1e-3


def test_bad_arguments_are_rejected(engine, frame40):
    fr = frame40
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    C = dsac_amd.capi
    for call in (lambda: engine.sample(4, max_tries=0),
                 lambda: engine.refine(np.zeros(6), np.zeros((1, 1600), np.int32), max_inl=1000),
                 lambda: engine.dPNP(np.zeros((2, 4), np.int32), eps=0.0),
                 lambda: engine.dRefine(np.zeros(6), np.zeros((1, 1600), np.int32), np.zeros(1600, np.int32), sub_sample=0.0)):
        with pytest.raises(C.DsacError) as ei:
            call()
        assert ei.value.code == C.DSAC_ERR_INVALID
    # quirk transpose needs a square map
    engine.set_frame(np.zeros((6, 3), np.float32), None, 2, 3, fr["cam"])
    with pytest.raises(C.DsacError):
        engine.dScore(np.zeros((1, 6)), np.zeros((1, 4), np.int32), np.zeros((1, 6), np.float32), quirk_transpose=True)
    # out-of-range set indices are clamped, never fault
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    p, s, ok = engine.sample(2, sets=np.array([[0, 1, 2, 10 ** 6], [-5, 3, 4, 5]], np.int32))
    assert p.shape == (2, 6)


def test_gather_rows(engine):
    """dsac_gather_rows: row i of dst = row rows[i] of src, one launch per 256 rows, 16-byte and 4-byte vector paths, repeated and out-of-order indices."""
    import torch
    from dsac_amd.capi import lib, ptr, check
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    for row_words, n_src, n_rows in ((4 * 1000, 7, 5), (3 * 333, 300, 700), (6, 40, 16)):
        src = torch.from_numpy(rng.integers(0, 1 << 30, size=(n_src, row_words), dtype=np.int32)).to(dev)
        rows = rng.integers(0, n_src, size=n_rows).astype(np.int32)
        dst = torch.zeros(n_rows, row_words, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        check(engine._ctx, lib.dsac_gather_rows(engine._ctx, ptr(dst), ptr(src), row_words * 4, n_rows, ptr(rows)))
        engine.synchronize()
        assert torch.equal(dst.cpu(), src.cpu()[torch.from_numpy(rows.astype(np.int64))])
    bad = np.array([0, -1], np.int32)
    with pytest.raises(Exception):
        check(engine._ctx, lib.dsac_gather_rows(engine._ctx, ptr(dst), ptr(src), 24, 2, ptr(bad)))
    with pytest.raises(Exception):
        check(engine._ctx, lib.dsac_gather_rows(engine._ctx, ptr(dst), ptr(src), 6, 1, ptr(rows)))  # not a multiple of 4


def test_loss_at_the_end_of_the_refinement_wave_equals_k7(engine, synth):
    """dsac_process_images computes maxLoss of a refined pose on the lane that holds it, at the end of K6's wave (csrc/loss_math.h; core/cnn_softam.h:1160-1179)
    instead of launching K7 behind the refinement: the four numbers equal K7 (dsac_loss_frames) on the refined poses bit for bit -- single frames with
    N a multiple of 64 and not, frame batches."""
    H, W = 120, 160
    perm = synth.fast_permutations(H * W, 8)
    for F, N in ((1, 256), (1, 100), (3, 128), (5, 256)):
        frames = [synth.chess_like_frame(H, W, seed=820 + f) for f in range(F)]
        xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
        uv, cam = frames[0]["uv"], frames[0]["cam"]
        gts = np.stack([fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]) for fr in frames])
        if F > 1:
            engine.set_frames(xyz, uv, H, W, cam)
        else:
            engine.set_frame(xyz[0], uv, H, W, cam)
        b = engine.processImages(N, perm, gt_jp6=gts, seed=9)
        assert (b["refSteps"] == 8).all() and np.isfinite(b["out4"]).all() and (b["out4"][:, 0] > 0).all()
        assert np.array_equal(engine.maxLossFrames(b["refAvgHyp"], gts)["out4"], b["out4"]), (F, N)


def test_misuse_of_the_seam_and_batch_entry_points_is_rejected(engine, frame40):
    """dsac_process_images_begin / _finish, dsac_softmax_frames, dsac_loss_batch_frames, dsac_select_frames, dsac_soft_score_derr: every misuse is an
    error code with a message, never a launch on bad sizes; the context stays usable afterwards."""
    import ctypes
    C = dsac_amd.capi
    lib, ptr = C.lib, C.ptr
    fr = frame40
    P = 1600
    ctx = engine._ctx
    xyz = np.stack([fr["xyz"], fr["xyz"]])
    engine.set_frames(xyz, fr["uv"], 40, 40, fr["cam"])
    perm = np.stack([np.random.default_rng(i).permutation(P).astype(np.int32) for i in range(8)])
    err = np.zeros((2 * 128, P), np.float32)

    def invalid(fn, code=C.DSAC_ERR_INVALID):
        with pytest.raises(C.DsacError) as ei:
            fn()
        assert ei.value.code == code and str(ei.value)

    invalid(lambda: engine.processImagesBegin(100, err))                      # a batch needs a multiple of 128 hypotheses per frame
    invalid(lambda: engine.processImagesBegin(128, err, max_tries=0))
    invalid(lambda: engine.processImagesBegin(0, err))
    # finish without a begin, and with a begin of another size
    sc = np.zeros(2 * 128)
    invalid(lambda: engine.processImagesFinish(128, sc, perm, np.zeros((256, 6))))
    poses, sets, ok = engine.processImagesBegin(128, err)
    invalid(lambda: engine.processImagesFinish(256, np.zeros(512), perm, np.zeros((512, 6))))
    invalid(lambda: engine.processImagesFinish(128, sc, perm, poses, max_inl=1000))
    # ... the open begin is still there: the matching finish goes through, a second finish does not
    r = engine.processImagesFinish(128, sc, perm, poses, gt_jp6=np.zeros((2, 6)))
    assert np.allclose(r["sfScores"].reshape(2, 128).sum(1), 1.0)
    invalid(lambda: engine.processImagesFinish(128, sc, perm, poses))

    w = np.zeros(256)
    ent = np.zeros(2)
    invalid(lambda: C.check(ctx, lib.dsac_softmax_frames(ctx, 0, 128, ptr(sc), 1.0, ptr(w), ptr(ent), None, None)))
    invalid(lambda: C.check(ctx, lib.dsac_softmax_frames(ctx, 2, 0, ptr(sc), 1.0, ptr(w), ptr(ent), None, None)))
    invalid(lambda: C.check(ctx, lib.dsac_softmax_frames(ctx, 2, 128, None, 1.0, ptr(w), ptr(ent), None, None)))

    est, gt, out4 = np.zeros((256, 6)), np.zeros((2, 6)), np.zeros((256, 4))
    invalid(lambda: C.check(ctx, lib.dsac_loss_batch_frames(ctx, 2, 0, ptr(est), ptr(gt), ptr(out4), None)))
    invalid(lambda: C.check(ctx, lib.dsac_loss_batch_frames(ctx, 2, 128, ptr(est), ptr(gt), None, None)))
    invalid(lambda: C.check(ctx, lib.dsac_loss_batch_frames(ctx, 2, 128, ptr(est), None, ptr(out4), None)))

    probs = np.full(256, 1.0 / 128)
    idx, el, g = np.zeros(2, np.int32), np.zeros(2), np.zeros(256)
    invalid(lambda: C.check(ctx, lib.dsac_select_frames(ctx, 2, 128, ptr(probs), ptr(out4), 4, ptr(np.array([0.5, 1.0])), ptr(idx), ptr(el), ptr(g))))
    invalid(lambda: C.check(ctx, lib.dsac_select_frames(ctx, 2, 128, ptr(probs), ptr(out4), 0, ptr(np.array([0.5, 0.5])), ptr(idx), ptr(el), ptr(g))))
    invalid(lambda: C.check(ctx, lib.dsac_select_frames(ctx, 0, 128, ptr(probs), ptr(out4), 4, ptr(np.array([0.5, 0.5])), ptr(idx), ptr(el), ptr(g))))
    C.check(ctx, lib.dsac_select_frames(ctx, 2, 128, ptr(probs), ptr(out4), 4, ptr(np.array([0.5, -1.0])), ptr(idx), ptr(el), ptr(g)))
    assert idx[0] == 64 and idx[1] == 0   # u = 0.5 of a uniform distribution: the first key above 64/128 (binary-exact sums); u < 0: the first maximum

    d_err = np.zeros_like(err)
    invalid(lambda: C.check(ctx, lib.dsac_soft_score_derr(ctx, 256, ptr(g), ptr(err), 100.0, 10.0, 0.0, ptr(d_err))))
    invalid(lambda: C.check(ctx, lib.dsac_soft_score_derr(ctx, -1, ptr(g), ptr(err), 100.0, 10.0, 0.5, ptr(d_err))))
    invalid(lambda: C.check(ctx, lib.dsac_soft_score_derr(ctx, 256, None, ptr(err), 100.0, 10.0, 0.5, ptr(d_err))))
    # refine_fd_sets on a batch: M must split into the frames, or every hypothesis names its frame
    sets3 = np.zeros((3, 4), np.int32)
    invalid(lambda: engine.dRefineSets(sets3, perm, np.zeros((3, P), np.int32)))
    # the context is still good
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    p, s, okk = engine.sample(8)
    assert okk.all()


def test_contexts_give_their_memory_back(frame40, synth):
    """dsac_destroy releases everything a context allocated (frame, staging slots, tail streams, events): forty create / use / destroy rounds, with a
    deferred tail pending at destruction in every other one, leave the device's free memory where it was."""
    import torch
    fr = frame40
    perm = synth.fast_permutations(1600, 8)

    def one_round(k):
        e = dsac_amd.Engine(0)
        if k & 1:
            e.set_option("pi_defer_tail", 2)
        xyz = np.stack([fr["xyz"]] * 3)
        e.set_frames(xyz, fr["uv"], 40, 40, fr["cam"])
        dev = torch.device("cuda", 0)
        out = dict(hyps=torch.zeros(384, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(384, 4, dtype=torch.int32, device=dev),
                   ok=torch.zeros(384, dtype=torch.uint8, device=dev), scores=torch.zeros(384, dtype=torch.float64, device=dev),
                   sfScores=torch.zeros(384, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(3, dtype=torch.float64, device=dev),
                   avgHyp=torch.zeros(3, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(3, 6, dtype=torch.float64, device=dev),
                   refSteps=torch.zeros(3, dtype=torch.int32, device=dev), out4=torch.zeros(3, 4, dtype=torch.float64, device=dev))
        e.processImages(128, torch.from_numpy(perm).to(dev), gt_jp6=torch.zeros(3, 6, dtype=torch.float64, device=dev), seed=k, out=out)
        e.close()  # with the tail of the call still in flight when k is odd
        torch.cuda.synchronize()
        return out["refSteps"].cpu().numpy()

    one_round(0)
    one_round(1)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(0)
    for k in range(40):
        sd = one_round(k)
        assert (sd == 8).all()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 <= 8 << 20, "device memory went from %d to %d bytes free over 40 contexts" % (free0, free1)

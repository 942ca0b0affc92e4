"""GPU parity of K6 (inlier refinement + finite-difference Jacobians) and K7 (pose loss) against the CPU oracle.

Both sides run the same fp64 arithmetic (lazy getDiffMap residuals, CvLevMarq state machine); differences are
libm-vs-ocml last bits, the summation order of the normal equations and the 6x6 solve (L D L^T on the GPU, Gaussian elimination in the oracle).
  refined poses : 1e-7 relative          inlier maps / step counts : identical
  dRefineHyp/Obj: central differences divide LM outputs by 2e-3 / 4, so 1e-4 relative to the largest entry
  loss, dLossMax: 1e-9
"""
import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu


def _forward(engine, orc, synth, fr, N=256, seed=3):
    engine.set_frame(fr["xyz"], fr["uv"], fr["H"], fr["W"], fr["cam"])
    poses, sets, ok, _ = orc.sample(N, seed, fr["xyz"], fr["uv"], fr["H"], fr["W"], fr["cam"])
    err = orc.get_diff_maps(poses, fr["xyz"], fr["uv"], fr["H"], fr["W"], fr["cam"])
    w = orc.softMax(0.1 * orc.soft_inlier(err, 10.0, 0.5))
    avg = orc.avg_pose(w, poses)
    perm = synth.fast_permutations(fr["H"] * fr["W"], 8, seed=5489)
    return avg, perm


def test_refine_forward_reference_size(engine, orc, synth, frame40):
    fr = frame40
    avg, perm = _forward(engine, orc, synth, fr)
    ref, imap_r, sd_r = orc.refine(avg, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"], want_inlier_map=True)
    got, sd, imap = engine.refine(avg, perm, want_inlier_map=True)
    assert np.array_equal(sd, sd_r) and sd[0] == 8
    assert np.array_equal(imap, imap_r)
    margin("a6", "K6 refine 40x40: refined pose vs oracle, max |d| / max(1, |pose|) (inlier map and step count identical)", np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), 1e-7)
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9)
    # the refined pose is close to the ground truth (sanity of the whole forward path)
    Re, te = orc.cv2our(got[0])
    Rg, tg = orc.cv2our(fr["gt_pose"])
    rot, tr = orc.pose_errors(Re, te, Rg, tg)
    assert rot < 1.0 and tr < 20.0


def test_refine_full_resolution(engine, orc, synth, frame_full):
    fr = frame_full
    avg, perm = _forward(engine, orc, synth, fr, N=64)
    ref, imap_r, sd_r = orc.refine(avg, perm, fr["xyz"], fr["uv"], fr["H"], fr["W"], fr["cam"], want_inlier_map=True)
    got, sd, imap = engine.refine(avg, perm, want_inlier_map=True)
    assert np.array_equal(sd, sd_r)
    assert np.array_equal(imap, imap_r)
    margin("a6", "K6 refine 640x480: refined pose vs oracle, max |d| / max(1, |pose|) (inlier map and step count identical)", np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), 1e-7)
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9)


def test_refine_replicas_and_perturbation(engine, orc, synth, frame40):
    fr = frame40
    avg, perm = _forward(engine, orc, synth, fr)
    rng = np.random.default_rng(0)
    B = 9
    init = avg[None, :] + rng.normal(scale=[1e-3] * 3 + [1.0] * 3, size=(B, 6))
    px = np.stack([rng.integers(-1, 1600, size=B), rng.integers(0, 3, size=B)], -1).astype(np.int32)
    px[0, 0] = -1
    val = (fr["xyz"][np.maximum(px[:, 0], 0), px[:, 1]] + rng.choice([-2.0, 2.0], size=B)).astype(np.float32)
    ref, sd_r = orc.refine(init, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"], pert_px_c=px, pert_value=val)
    got, sd = engine.refine(init, perm, pert_px_c=px, pert_value=val)
    assert np.array_equal(sd, sd_r)
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9)


def test_refine_too_few_inliers_and_short_perm(engine, orc, synth, frame40):
    fr = frame40
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    perm = synth.fast_permutations(1600, 3)
    bad = np.array([0.5, -0.3, 0.2, 100.0, 50.0, 900.0])  # far from the truth: < 50 inliers -> loop stops, pose unchanged
    got, sd, imap = engine.refine(bad, perm, want_inlier_map=True)
    ref, imap_r, sd_r = orc.refine(bad, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"], want_inlier_map=True)
    assert sd[0] == sd_r[0] == 0
    assert np.array_equal(got[0], bad)
    assert np.array_equal(imap, imap_r)
    # zero refinement steps requested
    got, sd = engine.refine(bad, perm[:0].reshape(0, 1600))
    assert np.array_equal(got[0], bad) and sd[0] == 0


@pytest.mark.parametrize("max_inl,min_inl,outliers,full", [(100, 50, 0.93, False), (256, 50, 0.3, False), (130, 129, 0.3, False),
                                                             (65, 1, 0.5, False), (100, 50, 0.9, True)])
def test_refine_sparse_inliers_and_other_inlier_counts(engine, orc, synth, max_inl, min_inl, outliers, full):
    """The walk reads the permutation 256 cells at a time and prefetches the next step's head; the LM loop takes two correspondences
    per lane and trip.  Sparse inliers make the walk span many batches (and stop inside one); max_inl = 256 / 130 / 65 give the LM
    loop 2 full trips / a ragged second trip / a second half with one lane.  `full`: 640x480 without sampled positions (the kernel
    derives the pixel from the cell index)."""
    H, W = (480, 640) if full else (120, 160)
    fr = synth.chess_like_frame(H, W, seed=77, outlier_frac=outliers)
    engine.set_frame(fr["xyz"], None if full else fr["uv"], H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8, seed=11)
    rng = np.random.default_rng(3)
    init = fr["gt_pose"][None, :] + rng.normal(size=(5, 6)) * np.array([0.004, 0.004, 0.004, 4.0, 4.0, 4.0])
    kw = dict(inlier_count=max_inl, min_inliers=min_inl)
    ref, sd_r = orc.refine(init, perm, fr["xyz"], fr["uv"], H, W, fr["cam"], **kw)
    got, sd = engine.refine(init, perm, max_inl=max_inl, min_inl=min_inl)
    assert np.array_equal(sd, sd_r) and sd.max() == 8
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9)
    _, imap_r, _ = orc.refine(init[0], perm, fr["xyz"], fr["uv"], H, W, fr["cam"], want_inlier_map=True, **kw)
    _, _, imap = engine.refine(init[0], perm, max_inl=max_inl, min_inl=min_inl, want_inlier_map=True)
    assert np.array_equal(imap, imap_r)
    if sd[0] == 8:
        assert imap.sum() == 8 * max_inl


def test_drefine_parity(engine, orc, synth, frame40):
    fr = frame40
    avg, perm = _forward(engine, orc, synth, fr)
    ref, imap, sd = orc.refine(avg, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"], want_inlier_map=True)
    Jh_r = orc.dRefineHyp(avg, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    Jo_r = orc.dRefineObj(avg, perm, imap, fr["xyz"], fr["uv"], 40, 40, fr["cam"], sub_sample=0.01)
    Jh, px, Jo = engine.dRefine(avg, perm, imap, sub_sample=0.01)
    margin("a14", "dRefineHyp (12 finite-difference replicas, one launch) vs oracle: max-rel", np.abs(Jh - Jh_r).max() / np.abs(Jh_r).max(), 1e-4)
    dense = np.zeros((6, 1600 * 3))
    for i, p in enumerate(px):
        dense[:, p * 3:p * 3 + 3] = Jo[i]
    nz = np.flatnonzero(np.abs(Jo_r).sum(0))
    assert set(nz // 3) <= set(int(p) for p in px) and len(px) >= 1  # a selected cell may have an exactly zero column (never walked)
    margin("a14", "dRefineObj (6 replicas per selected inlier cell) vs oracle: max-rel", np.abs(dense - Jo_r).max() / max(np.abs(Jo_r).max(), 1e-12), 1e-4)
    # denser sub-sampling exercises more replicas
    Jo_r2 = orc.dRefineObj(avg, perm, imap, fr["xyz"], fr["uv"], 40, 40, fr["cam"], sub_sample=0.1)
    _, px2, Jo2 = engine.dRefine(avg, perm, imap, sub_sample=0.1)
    dense2 = np.zeros((6, 1600 * 3))
    for i, p in enumerate(px2):
        dense2[:, p * 3:p * 3 + 3] = Jo2[i]
    assert len(px2) > len(px)
    assert np.abs(dense2 - Jo_r2).max() <= 1e-4 * max(np.abs(Jo_r2).max(), 1e-12)


def test_drefine_large_map_uses_the_tiled_plan(engine, orc, synth):
    """Maps above 16384 cells build the replica list with two tiled launches (64-column tiles, 16 row segments) instead of one workgroup:
    the selected cells (every skip-th inlier in the reference's column-major order, cnn_softam.h:868-880) and their Jacobians must be the
    oracle's.  160 columns = two full tiles and a half one, 120 rows = 15 used row segments of 8."""
    H, W = 120, 160
    fr = synth.chess_like_frame(H, W, seed=5)
    engine.set_frame(fr["xyz"], fr["uv"], H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8, seed=3)
    init = fr["gt_pose"] + np.array([0.003, -0.002, 0.001, 2.0, -3.0, 4.0])
    ref, imap, sd = orc.refine(init, perm, fr["xyz"], fr["uv"], H, W, fr["cam"], want_inlier_map=True)
    assert sd[0] == 8 and (imap > 0).sum() > 300
    for sub in (0.05, 0.3):
        Jh_r = orc.dRefineHyp(init, perm, fr["xyz"], fr["uv"], H, W, fr["cam"])
        Jo_r = orc.dRefineObj(init, perm, imap, fr["xyz"], fr["uv"], H, W, fr["cam"], sub_sample=sub)
        Jh, px, Jo = engine.dRefine(init, perm, imap, sub_sample=sub)
        # the reference's selection: walk x outer / y inner, every skip-th inlier
        skip = int(1 / sub)
        order = [y * W + x for x in range(W) for y in range(H) if imap[y * W + x] > 0]
        want = order[skip - 1::skip]
        assert [int(p_) for p_ in px] == want
        assert np.abs(Jh - Jh_r).max() <= 1e-4 * max(np.abs(Jh_r).max(), 1e-12)
        dense = np.zeros((6, H * W * 3))
        for i, p_ in enumerate(px):
            dense[:, p_ * 3:p_ * 3 + 3] = Jo[i]
        assert np.abs(dense - Jo_r).max() <= 1e-4 * max(np.abs(Jo_r).max(), 1e-12)


def test_loss_and_gradient(engine, orc, synth):
    rng = np.random.default_rng(1)
    gt_cv = np.array([0.2, -0.1, 0.05, 120.0, -340.0, 2100.0])
    gt_jp = orc.cv_to_jp6(gt_cv)
    Rg, tg = orc.cv2our(gt_cv)
    cases = [gt_cv + np.array([0.05, 0.02, -0.03, 1.0, 2.0, -1.0]),      # rotation error dominates
             gt_cv + np.array([1e-4, 0, 0, 80.0, -60.0, 150.0]),          # translation error dominates
             gt_cv.copy(),                                                  # zero error -> zero gradient
             gt_cv + rng.normal(scale=[0.3] * 3 + [500.0] * 3)]
    for est in cases:
        Re, te = orc.cv2our(est)
        r = engine.maxLoss(est, gt_jp, want_grad=True)
        rot, tr = orc.pose_errors(Rg, tg, Re, te)
        # acos near 1 turns 1e-16 of round-off in the trace into ~1e-6 deg, hence the absolute floor
        margin("a7", "K7 maxLoss vs oracle: |loss - oracle| (absolute floor: acos near 1 turns 1e-16 in the trace into 1e-6 deg)", abs(r["loss"] - orc.maxLoss(Rg, tg, Re, te)), 1e-5 + 1e-9 * r["loss"], stated=1e-9)
        assert abs(r["rotErr"] - rot) <= 1e-5 and abs(r["tErr"] - tr) <= 1e-7 * max(1.0, tr)
        assert r["correct"] == (rot < 5 and tr < 50)
        assert np.all(np.isfinite(r["grad"]))
        if rot + tr > 1e-3:  # at exactly zero error the gradient is 0/0 on both sides
            Jr = orc.dLossMax(orc.cv_to_jp6(est), gt_jp)
            margin("a8", "K7 dLossMax vs oracle: max-rel", np.abs(r["grad"] - Jr).max() / max(1.0, np.abs(Jr).max()), 1e-8)


def test_many_long_walks_as_two_launches_per_step_equal_the_fused_kernel(engine, synth):
    """Round 6 (core/cnn.h:1154-1230: the DSAC variant refines EVERY hypothesis): from 32 problems on a map of >= 16 384 cells a refinement step runs as two
    launches -- k_refine_walk (a scan of the step's permuted cells in chunks: every chunk's inliers in permutation order into HBM) and k_refine_lm (one wave
    per problem: the first max_inl inliers in chunk order, then the LM solve) -- instead of the fused kernel.  Same arithmetic, same lists: refined poses, step
    counts and per-problem inlier maps are bit-identical to the fused kernel with one ("k6_waves" 1) and with four waves per problem."""
    H, W = 120, 160
    P = H * W
    fr = synth.chess_like_frame(H, W, seed=41, outlier_frac=0.5, grid_uv=True)
    engine.set_frame(fr["xyz"], None, H, W, fr["cam"])
    perm = synth.fast_permutations(P, 8)
    rng = np.random.default_rng(3)
    B = 96
    scale = np.where(np.arange(B)[:, None] % 3 == 0, 1.0, 25.0)  # a third near the pose (walks that stop early), the rest far off (walks over the whole map)
    init = fr["gt_pose"][None, :] + rng.normal(size=(B, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0]) * scale
    res = {}
    try:
        for waves in (0, 1, 4):
            engine.set_option("k6_waves", waves)
            res[waves] = engine.refineAll(init, perm, max_inl=100, min_inl=50, thr=10.0, want_inlier_maps=True)
    finally:
        engine.set_option("k6_waves", 0)
    p0, s0, m0 = res[0]
    assert 0 < int((s0 == 8).sum()) < B or int(s0.max()) > 0  # a mix of finished and aborted refinements
    for waves in (1, 4):
        p, s, m = res[waves]
        assert np.array_equal(p0, p) and np.array_equal(s0, s) and np.array_equal(m0, m), "k6_waves %d" % waves
    # the scan's shapes (results never depend on them): 1 / 2 / 4 problems per wave, chunks of 256 ... 4096 cells (one round ... a chunk longer than a wave's
    # share of this map), the skip of chunks behind max_inl finished inliers off
    try:
        for tune in (1, 2, 4, 1 | 256 << 8, 4 | 256 << 8, 2 | 1024 << 8, 4 | 4096 << 8, 1 | 1 << 24, 4 | 512 << 8 | 1 << 24):
            engine.set_option("k6_scan_tune", tune)
            p, s, m = engine.refineAll(init, perm, max_inl=100, min_inl=50, thr=10.0, want_inlier_maps=True)
            assert np.array_equal(p0, p) and np.array_equal(s0, s) and np.array_equal(m0, m), "k6_scan_tune %#x" % tune
        with pytest.raises(Exception):
            engine.set_option("k6_scan_tune", 3)
        with pytest.raises(Exception):
            engine.set_option("k6_scan_tune", 1 | 100 << 8)
    finally:
        engine.set_option("k6_scan_tune", 0)
    # index lists that are no permutations (duplicates, a part of the map only): the scan's frame bounds are taken over the frame's cells, not the lists
    perm2 = ((perm.astype(np.int64) * 7) % (P // 3)).astype(np.int32)
    try:
        res2 = {}
        for waves in (0, 1):
            engine.set_option("k6_waves", waves)
            res2[waves] = engine.refineAll(init, perm2, max_inl=100, min_inl=50, thr=10.0, want_inlier_maps=True)
    finally:
        engine.set_option("k6_waves", 0)
    for a, b in zip(res2[0], res2[1]):
        assert np.array_equal(a, b)


def test_the_walks_fp32_filter_never_changes_a_decision(engine, orc, synth):
    """k_refine_walk discards a cell by an fp32 evaluation when that shows it further from the threshold than a proven error bound, and decides by the reference's fp64
    arithmetic otherwise.  A map built AGAINST the filter: for each of 64 poses ~290 cells whose reprojection lands at thr (1 +- eps), eps from 0 to 1e-3 (the float
    rounding of the coordinate scatters them a further ~1e-4 px to either side -- thousands of decisions within 1e-7 ... 1e-3 px of the threshold), and cells 1 um ...
    1 mm from the camera plane, on it and behind it.  Poses, step counts and every problem's inlier map must equal the fused kernel's (one wave per problem, fp64
    for every cell) and the walk with the filter switched off; four problems are also checked against the oracle's inlier maps."""
    H, W = 128, 160
    P = H * W
    fx, fy, cx, cy = synth.CAM_7SCENES
    rng = np.random.default_rng(2026)
    B, per = 64, 300
    uv = synth.pixel_grid(H, W, H, W).astype(np.float64)
    xyz = np.zeros((P, 3), np.float32)
    base = synth.chess_like_frame(H, W, seed=9, grid_uv=True)
    poses = base["gt_pose"][None, :] + rng.normal(size=(B, 6)) * np.array([0.05, 0.05, 0.05, 60.0, 60.0, 60.0])
    cells = rng.permutation(P)
    thr = 10.0
    n_close = 0
    for b in range(B):
        idx = cells[b * per:(b + 1) * per]
        R, t = synth.rodrigues(poses[b, :3]), poses[b, 3:]
        eps = rng.choice([0.0, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3], size=per) * rng.choice([-1.0, 1.0], size=per)
        th = rng.uniform(0, 2 * np.pi, size=per)
        depth = rng.uniform(300.0, 4000.0, size=per)
        # the last 12 cells of every group: at, next to and behind the camera plane
        depth[-12:] = np.array([0.0, 1e-3, -1e-3, 1e-2, 0.1, 1.0, -1.0, 1e-4, -1e-2, 3e-3, 0.5, -0.5])
        u = uv[idx, 0] + thr * (1 + eps) * np.cos(th)
        v = uv[idx, 1] + thr * (1 + eps) * np.sin(th)
        Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], -1)
        xyz[idx] = ((Xc - t) @ R).astype(np.float32)
        e = orc.get_diff_maps(poses[b], xyz[idx], uv[idx].astype(np.float32), 1, per, synth.CAM_7SCENES)[0]
        n_close += int((np.abs(e[:-12] - thr) < 1e-3).sum())
    rest = cells[B * per:]
    xyz[rest] = base["xyz"][rest]
    assert n_close > 0.5 * B * (per - 12), n_close  # the construction works: most built cells are within 1e-3 px of the threshold in the reference's arithmetic
    engine.set_frame(xyz, None, H, W, synth.CAM_7SCENES)
    perm = synth.fast_permutations(P, 8, seed=77)
    res = {}
    try:
        for name, waves, exact in (("walk", 0, 0), ("fused", 1, 0), ("walk-exact", 0, 1)):
            engine.set_option("k6_waves", waves)
            engine.set_option("k6_walk_exact", exact)
            res[name] = engine.refineAll(poses, perm, max_inl=100, min_inl=50, thr=thr, want_inlier_maps=True)
    finally:
        engine.set_option("k6_waves", 0)
        engine.set_option("k6_walk_exact", 0)
    p0, s0, m0 = res["walk"]
    assert int(m0.sum()) > 0
    for name in ("fused", "walk-exact"):
        p, s, m = res[name]
        assert np.array_equal(s0, s) and np.array_equal(m0, m) and np.array_equal(p0, p), name
    uvf = uv.astype(np.float32)
    for b in (0, 21, 42, 63):
        _, imap_r, sd_r = orc.refine(poses[b], perm, xyz, uvf, H, W, synth.CAM_7SCENES, want_inlier_map=True, inlier_count=100, min_inliers=50)
        assert sd_r[0] == s0[b] and np.array_equal(np.asarray(imap_r).reshape(-1), m0[b].reshape(-1)), b


@pytest.mark.parametrize("case", ["three frames x 174 (two problems per wave)", "three frames x 175 (one per wave)", "NaN / inf cells", "threshold beyond the clamp",
                                  "256 inliers per step", "per-frame pixel positions"])
def test_scan_corner_cases_equal_the_fused_kernel(engine, synth, case):
    """The scan's less travelled branches against the fused kernel (k6_waves 1), bit for bit: frame batches whose hypotheses per frame are no multiple of four
    (>= 512 problems: two problems per wave, or one), NaN and infinite coordinates (the frame bound becomes NaN / inf: every cell takes the fp64 residual),
    thresholds at or beyond the error clamp (every cell an inlier: the fp64 path by rule), max_inl = 256 (the LM kernel's LDS capacity), sampled pixel positions
    per frame."""
    H, W = 128, 160
    P = H * W
    rng = np.random.default_rng(12)
    F, per, kw, own_uv = 1, 64, dict(max_inl=100, min_inl=50, thr=10.0), False
    if case.startswith("three frames x 174"): F, per = 3, 174
    if case.startswith("three frames x 175"): F, per = 3, 175
    if case == "threshold beyond the clamp": kw = dict(max_inl=100, min_inl=50, thr=120.0)
    if case == "256 inliers per step": kw = dict(max_inl=256, min_inl=50, thr=10.0)
    if case == "per-frame pixel positions": F, per, own_uv = 2, 32, True
    frames = [synth.chess_like_frame(H, W, seed=300 + f, outlier_frac=0.6, grid_uv=not own_uv) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    if case == "NaN / inf cells":
        xyz[0, rng.choice(P, 50, replace=False)] = np.nan
        xyz[0, rng.choice(P, 50, replace=False), 0] = np.inf
    uv = np.ascontiguousarray(np.stack([fr["uv"] for fr in frames])) if own_uv else None
    engine.set_frames(xyz, uv, H, W, frames[0]["cam"], uv_per_frame=own_uv)
    perm = synth.fast_permutations(P, 8, seed=5)
    scale = np.where(np.arange(F * per)[:, None] % 4 == 0, 1.0, 20.0)
    init = np.concatenate([np.repeat(fr["gt_pose"][None, :], per, 0) for fr in frames]) + rng.normal(size=(F * per, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0]) * scale
    res = {}
    try:
        for waves in (0, 1):
            engine.set_option("k6_waves", waves)
            res[waves] = engine.refineAll(init, perm, want_inlier_maps=True, **kw)
    finally:
        engine.set_option("k6_waves", 0)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b, equal_nan=True), case
    assert int(res[0][1].max()) > 0

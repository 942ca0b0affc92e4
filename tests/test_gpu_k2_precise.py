"""K2's PRECISE mode (round 5; dsac_set_option("k2_flags", 1 << 25)): the reference's own arithmetic -- the projection in double, one rounding to float
of each image-plane difference (core/cnn_softam.h:319-362) -- instead of the fp32 matrix-core transform.  What it buys, against the oracle at the TIMED
shapes: residuals within 2e-4 px (the fast forms: 5.9e-4 of the stated 1e-3) and softmax weights within the stated 1e-4 also in a tie,
where the fast forms are asserted at 1e-3 only (tests/test_gpu_timed_configs.py)."""
import numpy as np
import pytest

from conftest import excl_clamp_edge, margin

pytestmark = pytest.mark.gpu

H, W = 480, 640
P = H * W
TAU, BETA, SCALE, CLAMP = 10.0, 0.5, 0.1, 100.0
PRECISE = 1 << 25


@pytest.fixture()
def precise_engine(engine):
    engine.set_option("k2_variant", -1)
    engine.set_option("k2_flags", PRECISE)
    yield engine
    engine.set_option("k2_flags", 0)


def test_residuals_and_tied_weights_at_640x480(precise_engine, orc, synth):
    eng = precise_engine
    fr = synth.chess_like_frame(H, W, seed=1305 + 1000)
    uv, cam = synth.pixel_grid(H, W), fr["cam"]
    eng.set_frame(fr["xyz"], None, H, W, cam)
    poses, sets, ok = eng.sample(256, seed=4711, thr=10.0, max_tries=1 << 16)
    err, soft = np.zeros((256, P), np.float32), np.zeros(256)
    eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
    ref = orc.get_diff_maps(poses, fr["xyz"], uv, H, W, cam)
    m = excl_clamp_edge(err, ref, CLAMP)
    margin("a3", "K2 PRECISE mode, residuals at 640x480 x 256 hypotheses (all cells): max |err - oracle| px", np.abs(err - ref)[m].max(), 2e-4, stated=1e-3)
    # the fast form on the same poses, for the record (asserted at 1e-3 elsewhere)
    eng.set_option("k2_flags", 0)
    eng.set_option("k2_exact_auto", 0)  # round 6: the auto policy would take the exact form
    err_f, soft_f = np.zeros((256, P), np.float32), np.zeros(256)
    eng.reproject(poses, err=err_f, soft=soft_f, tau=TAU, beta=BETA)
    eng.set_option("k2_exact_auto", 1)
    eng.set_option("k2_flags", PRECISE)
    mf = excl_clamp_edge(err_f, ref, CLAMP)
    fast = np.abs(err_f - ref)[mf].max()
    print("fast form on the same poses: %.2e px; precise: %.2e px" % (fast, np.abs(err - ref)[m].max()))
    # Where the fast (fp32 matrix-core) form's error lives: it is an ABSOLUTE error of the camera-frame point (four fp32 roundings at the magnitude of
    # f R.X ~ 1e6 and of z ~ 3e3) divided by the depth, so over ALL 78 M cells of a frame -- the suite's other tests sample rows -- a handful of cells that a
    # hypothesis places within a few hundred millimetres of its camera centre exceed the stated 1e-3 px.  Recorded here with the bound it obeys:
    # |err - oracle| <= max(1e-3, 0.5 / depth[mm]) px, depth = the cell's camera-frame z under the hypothesis.
    d = np.abs(err_f - ref)
    d[~mf] = 0
    over = np.argwhere(d > 1e-3)
    worst_ratio, n_over = 0.0, len(over)
    for h in np.unique(over[:, 0]) if n_over else []:
        R = synth.rodrigues(poses[h, :3])
        cells = over[over[:, 0] == h, 1]
        z = np.abs(fr["xyz"][cells].astype(np.float64) @ R[2] + poses[h, 5])
        worst_ratio = max(worst_ratio, float((d[h, cells] * z).max()))
    print("fast form: %d of %d cells above 1e-3 px (%.2e of them), max error x depth = %.3f px mm" % (n_over, d.size, n_over / d.size, worst_ratio))
    margin("a3", "K2 fast form over ALL cells of 256 x 640x480: fraction of cells above the stated 1e-3 px (cells within ~0.5 m of the camera centre)", n_over / d.size, 1e-5)
    margin("a3", "K2 fast form over ALL cells: max of |err - oracle| x depth over the cells above 1e-3 px [px mm] (error <= 0.5 / depth)", worst_ratio, 0.5)
    # scores: 307 200 sigmoids each.  The weights of two hypotheses in a tie move by w (1 - w) * scale * (error of their score DIFFERENCE) <= 0.25 * 0.1 * 2 max|d score|
    soft_ref = orc.soft_inlier(ref, TAU, BETA)
    dsv = soft - soft_ref
    ds = np.abs(dsv).max()
    margin("north*", "K2 PRECISE mode: soft-inlier scores at 640x480, max |score - oracle| relative to the largest score", ds / soft_ref.max(), 1e-6, stated=1e-4)
    print("max |score - oracle|: precise %.2e, fast %.2e (scores up to %.0f)" % (ds, np.abs(soft_f - soft_ref).max(), soft_ref.max()))
    # The weights of two hypotheses in a TIE move by w (1 - w) * scale * (error of their score DIFFERENCE) = 0.25 * 0.1 * |d_i - d_j|.  Part of a score's
    # error is systematic (the hardware exp2 / rcp are not exactly rounded: a bias that grows with the score) and cancels between two hypotheses whose scores
    # tie; what does not cancel is measured on every pair of UNRELATED hypotheses whose oracle scores lie within 5 % of the top score of each other
    order = np.argsort(-soft_ref)
    worst_pair = 0.0
    npairs = 0
    for a in range(len(order)):
        for b in range(a + 1, len(order)):
            i, j = order[a], order[b]
            if soft_ref[i] - soft_ref[j] > 0.05 * soft_ref.max():
                break
            worst_pair = max(worst_pair, abs(dsv[i] - dsv[j]))
            npairs += 1
    assert npairs >= 100
    margin("a4", "K2 PRECISE mode: softmax-weight error in a tie of two UNRELATED hypotheses, scale 0.1 -- 0.25 x scale x max |d_i - d_j| over %d near-tie pairs" % npairs,
           0.25 * SCALE * worst_pair, 1e-4)
    dsf = soft_f - soft_ref
    worst_fast = max(abs(dsf[order[a]] - dsf[order[b]]) for a in range(len(order)) for b in range(a + 1, len(order)) if soft_ref[order[a]] - soft_ref[order[b]] <= 0.05 * soft_ref.max())
    print("near-tie pairs: precise %.2e, fast %.2e (0.25 x scale x max |d_i - d_j|)" % (0.25 * SCALE * worst_pair, 0.25 * SCALE * worst_fast))
    # Why the fast form's scores are off by ~0.1: diagnostic build of the precise kernel with ONLY the pose records rounded to float (k2_flags bit 26) --
    # everything else exact.  A rounded record is a tiny fixed perturbation of the hypothesis' pose: it moves ALL of its projections the same way, so the
    # error of the score does not average out over the cells as independent per-cell rounding does
    eng.set_option("k2_flags", PRECISE | (1 << 26))
    soft_r = np.zeros(256)
    eng.reproject(poses, soft=soft_r, tau=TAU, beta=BETA)
    eng.set_option("k2_flags", PRECISE)
    dsr = soft_r - soft_ref
    pair_r = max(abs(dsr[order[a]] - dsr[order[b]]) for a in range(len(order)) for b in range(a + 1, len(order)) if soft_ref[order[a]] - soft_ref[order[b]] <= 0.05 * soft_ref.max())
    print("records rounded to fp32, everything else exact: max |score - oracle| %.2e, near-tie pairs %.2e (fast form: %.2e / %.2e; precise: %.2e / %.2e)" %
          (np.abs(dsr).max(), 0.25 * SCALE * pair_r, np.abs(dsf).max(), 0.25 * SCALE * worst_fast, ds, 0.25 * SCALE * worst_pair))
    margin("a4", "K2 diagnostic: share of the fast form's near-tie weight error that the fp32 pose RECORD alone reproduces (records rounded, rest exact)",
           (0.25 * SCALE * pair_r) / (0.25 * SCALE * worst_fast), 0.2, at_least=True)
    # ... and the constructed tie of tests/test_gpu_timed_configs.py at the STATED tolerance
    best = int(np.argmax(soft))
    tie = poses.copy()
    tie[(best + 1) % 256] = poses[best] + np.array([1e-9, -1e-9, 1e-9, 1e-6, 1e-6, -1e-6])
    s2 = eng.softInlierScores(tie, tau=TAU, beta=BETA)
    w, _, _ = eng.softMax(s2, SCALE)
    s2_ref = orc.soft_inlier(orc.get_diff_maps(tie, fr["xyz"], uv, H, W, cam), TAU, BETA)
    w_ref = orc.softMax(SCALE * s2_ref)
    assert np.sort(w_ref)[-2] > 0.2
    margin("a4", "K2 PRECISE mode: softmax weights at 640x480 in a TIE of the two best hypotheses, scale 0.1: max |w - oracle|", np.abs(w - w_ref).max(), 1e-4)


def test_bench_shape_and_every_entry_point(precise_engine, orc, synth):
    """16 frames x 256 hypotheses through dsac_score_hypotheses_frames, dsac_process_images and the begin / finish pair with the flag set: one kernel, the
    same error images bit for bit; sampled rows within 2e-4 px of the oracle."""
    import torch
    eng = precise_engine
    dev = torch.device("cuda", 0)
    F, N = 16, 256
    frames = [synth.chess_like_frame(H, W, seed=1305 + 1000 + f) for f in range(F)]
    xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
    cam, uv = frames[0]["cam"], synth.pixel_grid(H, W)
    eng.set_frames(xyz, None, H, W, cam, borrow=True)
    f64 = dict(dtype=torch.float64, device=dev)
    err = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    out = (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev), torch.zeros(F * N, **f64),
           torch.zeros(F * N, **f64), torch.zeros(F, **f64), torch.zeros(F, 6, **f64))
    eng.profile_enable(True, stride=1)
    eng.scoreHypothesesFrames(N, seed=4711, thr=10.0, max_tries=1 << 16, err=err, out=out)
    eng.synchronize()
    ms, n = eng.profile_read(0, reset=True)
    eng.profile_enable(False)
    ph, sf = out[0].cpu().numpy(), out[3].cpu().numpy()
    rng = np.random.default_rng(0)
    worst = 0.0
    for f in range(0, F, 3):
        rows = f * N + rng.choice(N, 6, replace=False)
        got = err[torch.as_tensor(rows, device=dev)].cpu().numpy()
        ref = orc.get_diff_maps(ph[rows], frames[f]["xyz"], uv, H, W, cam)
        m = excl_clamp_edge(got, ref, CLAMP)
        worst = max(worst, np.abs(got - ref)[m].max())
    margin("a3", "K2 PRECISE mode at the bench shape (16 x 256 x 640x480, sampled rows): max |err - oracle| px", worst, 2e-4, stated=1e-3)
    print("precise K2 at the bench shape: %.1f us per launch" % (ms * 1e3))
    # the other entry points launch the same kernel on the same poses
    err2 = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    soft2 = torch.zeros(F * N, **f64)
    p2 = (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev))
    eng.processImagesBegin(N, err2, seed=4711, thr=10.0, max_tries=1 << 16, soft=soft2, out=p2)
    eng.synchronize()
    assert torch.equal(p2[0], out[0]) and torch.equal(err2, err) and torch.equal(soft2, out[3])
    err2.zero_()
    torch.cuda.synchronize()  # the fill ran on torch's stream, the engine has its own
    eng.reproject(out[0], N=F * N, err=err2)
    eng.synchronize()
    assert torch.equal(err2, err)


def test_records_in_two_pieces(engine, orc, synth):
    """k2_flags bit 27: the fast matrix-core form with the low parts of the pose records through fp16 matrix-core instructions, as the accumulator the fp32
    products are added onto.  Measured: the score error 0.112 -> 0.017 and the tie of unrelated hypotheses 4.4e-3 -> 7.2e-4 (85 % of the systematic error
    gone) for +16 % of K2's time -- a middle mode; it does NOT reach the stated 1e-4 (what is left is the matrix core's own fp32 accumulation), the precise
    mode (bit 25) does."""
    RECLO = 1 << 27
    fr = synth.chess_like_frame(H, W, seed=1305 + 1000)
    uv, cam = synth.pixel_grid(H, W), fr["cam"]
    engine.set_option("k2_variant", -1)
    engine.set_option("k2_flags", 0)
    engine.set_frame(fr["xyz"], None, H, W, cam)
    poses, sets, ok = engine.sample(256, seed=4711, thr=10.0, max_tries=1 << 16)
    ref = orc.get_diff_maps(poses, fr["xyz"], uv, H, W, cam)
    soft_ref = orc.soft_inlier(ref, TAU, BETA)
    order = np.argsort(-soft_ref)
    pairs = [(order[a], order[b]) for a in range(len(order)) for b in range(a + 1, len(order)) if soft_ref[order[a]] - soft_ref[order[b]] <= 0.05 * soft_ref.max()]
    try:
        res = {}
        for name, flags, var in (("fast", 0, -1), ("two-piece records", RECLO, -1), ("two-piece records, 3 waves per SIMD", RECLO, 80)):
            engine.set_option("k2_variant", var)
            engine.set_option("k2_flags", flags)
            engine.set_option("k2_exact_auto", 0 if name == "fast" else 1)  # round 6: the auto policy's default is the exact form; "fast" = the fp32 matrix-core form
            err, soft = np.zeros((256, P), np.float32), np.zeros(256)
            engine.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
            m = excl_clamp_edge(err, ref, CLAMP)
            d = soft - soft_ref
            res[name] = (np.abs(err - ref)[m].max(), np.abs(d).max(), 0.25 * SCALE * max(abs(d[i] - d[j]) for i, j in pairs), float(np.mean(np.abs(err - ref)[m])))
            print("%-40s max |err - oracle| %.2e px (mean %.2e), max |score - oracle| %.2e, near-tie weight error %.2e" % ((name,) + res[name][:1] + res[name][3:] + res[name][1:3]))
        r = res["two-piece records"]
        margin("a3", "K2 with two-piece records (k2_flags bit 27): residuals over all cells, max |err - oracle| px (the accumulation's rare cells stay)", r[0], 6e-3, stated=1e-3)
        margin("north*", "K2 with two-piece records: max |score - oracle| relative to the largest score", r[1] / soft_ref.max(), 3e-7, stated=1e-4)
        margin("a4", "K2 with two-piece records: softmax-weight error in a tie of two UNRELATED hypotheses (near-tie pairs), scale 0.1", r[2], 1.5e-3, stated=1e-4)
        assert r[2] < 0.25 * res["fast"][2], "the correction did not remove the systematic part of the error"
        assert abs(res["two-piece records, 3 waves per SIMD"][2] - r[2]) < 1e-12  # same arithmetic, other occupancy
    finally:
        engine.set_option("k2_flags", 0)
        engine.set_option("k2_variant", -1)
        engine.set_option("k2_exact_auto", 1)

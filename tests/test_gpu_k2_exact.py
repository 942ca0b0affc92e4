"""K2's EXACT-TRANSFORM form (round 6; dsac_set_option("k2_flags", 1 << 28)).  The reference projects in double and rounds each image-plane difference to
float once (core/cnn_softam.h:319-362).  This form evaluates E = R.X + t from fixed-point fp16 pieces on the fp16 matrix core -- one accumulation per row is
exact, the other stays below a few millimetres -- rounds the camera-frame point to float once and polishes the hardware reciprocal with one Newton step
(dsac_amd/csrc/k_forward.hip: k_pose_prep_split, hp_chunk_ex).  What it must hold, against the oracle: NO cell above the stated 1e-3 px over ALL cells of a
frame, softmax weights of unrelated near-tie pairs within the stated 1e-4, at the bench shape ALL rows of a frame -- at the speed of the streaming forms
(profiles/r06_k2_exact_ab.txt), not the precise mode's."""
import numpy as np
import pytest

from conftest import excl_clamp_edge, margin

pytestmark = pytest.mark.gpu

H, W = 480, 640
P = H * W
TAU, BETA, SCALE, CLAMP = 10.0, 0.5, 0.1, 100.0
EXACT = 1 << 28


@pytest.fixture()
def exact_engine(engine):
    engine.set_option("k2_variant", -1)
    engine.set_option("k2_flags", EXACT)
    yield engine
    engine.set_option("k2_flags", 0)
    engine.set_option("k2_variant", -1)
    engine.set_option("k2_exact_auto", 1)


def test_the_default_policy_is_the_exact_form(engine, orc, synth):
    """Round 6: with no option set the auto policy launches the exact-transform form wherever it applies (vectorisable map, focal length <= 1024) -- the
    default K2 holds every stated tolerance; k2_exact_auto = 0 gives the fp32 matrix-core forms of rounds 2-5 back.  An arithmetic form that is ASKED for and
    cannot run (a map the vector kernels cannot read) is an error, not a silent fp32 launch."""
    import dsac_amd
    fr = synth.chess_like_frame(H, W, seed=2305)
    engine.set_option("k2_variant", -1)
    engine.set_option("k2_flags", 0)
    engine.set_option("k2_exact_auto", 1)
    engine.set_frame(fr["xyz"], None, H, W, fr["cam"])
    poses, _, _ = engine.sample(128, seed=4711, thr=10.0, max_tries=1 << 16)
    e_def, e_ex, e_fast = (np.zeros((128, P), np.float32) for _ in range(3))
    engine.reproject(poses, err=e_def)
    engine.set_option("k2_flags", EXACT)
    engine.reproject(poses, err=e_ex)
    engine.set_option("k2_flags", 0)
    engine.set_option("k2_exact_auto", 0)
    engine.reproject(poses, err=e_fast)
    engine.set_option("k2_exact_auto", 1)
    assert np.array_equal(e_def, e_ex) and not np.array_equal(e_def, e_fast)
    # 53 x 37: H*W is odd -- no 16-byte vectors
    fo = synth.chess_like_frame(37, 53, seed=5, grid_uv=True)
    engine.set_frame(fo["xyz"], None, 37, 53, fo["cam"])
    po, _, _ = engine.sample(64, seed=1, thr=10.0, max_tries=1 << 16)
    eo = np.zeros((64, 37 * 53), np.float32)
    engine.reproject(po, err=eo)  # the default falls back quietly
    ref = orc.get_diff_maps(po, fo["xyz"], fo["uv"], 37, 53, fo["cam"])
    m = excl_clamp_edge(eo, ref, CLAMP)
    assert np.abs(eo - ref)[m].max() <= 1e-3
    for flag in (EXACT, 1 << 25, 1 << 27):
        engine.set_option("k2_flags", flag)
        with pytest.raises(dsac_amd.capi.DsacError):
            engine.reproject(po, err=eo)
    engine.set_option("k2_flags", 0)


def near_tie_pairs(soft_ref):
    order = np.argsort(-soft_ref)
    return [(order[a], order[b]) for a in range(len(order)) for b in range(a + 1, len(order)) if soft_ref[order[a]] - soft_ref[order[b]] <= 0.05 * soft_ref.max()]


@pytest.mark.parametrize("seed", [2305, 2306, 2307])
def test_all_cells_and_unrelated_ties_at_640x480(exact_engine, orc, synth, seed):
    eng = exact_engine
    fr = synth.chess_like_frame(H, W, seed=seed)
    uv, cam = synth.pixel_grid(H, W), fr["cam"]
    eng.set_frame(fr["xyz"], None, H, W, cam)
    poses, sets, ok = eng.sample(256, seed=4711, thr=10.0, max_tries=1 << 16)
    err, soft = np.zeros((256, P), np.float32), np.zeros(256)
    eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
    ref = orc.get_diff_maps(poses, fr["xyz"], uv, H, W, cam)
    m = excl_clamp_edge(err, ref, CLAMP)
    d = np.abs(err - ref)
    d[~m] = 0
    margin("a3", "K2 EXACT form, residuals over ALL cells of 256 x 640x480: max |err - oracle| px", d.max(), 1e-3)
    assert int((d > 1e-3).sum()) == 0
    margin("a3", "K2 EXACT form: mean |err - oracle| px over all cells (the float rounding of the camera-frame point)", d[m].mean(), 2e-5)
    soft_ref = orc.soft_inlier(ref, TAU, BETA)
    dsv = soft - soft_ref
    margin("north*", "K2 EXACT form: soft-inlier scores, max |score - oracle| relative to the largest score", np.abs(dsv).max() / soft_ref.max(), 2e-7, stated=1e-4)
    pairs = near_tie_pairs(soft_ref)
    assert len(pairs) >= 100
    tie = 0.25 * SCALE * max(abs(dsv[i] - dsv[j]) for i, j in pairs)
    margin("a4", "K2 EXACT form: softmax-weight error in a tie of two UNRELATED hypotheses, scale 0.1 -- 0.25 x scale x max |d_i - d_j| over the near-tie pairs", tie, 1e-4)
    # the fast form on the same poses: what the exact form buys
    eng.set_option("k2_flags", 0)
    eng.set_option("k2_exact_auto", 0)  # the auto policy takes the exact form by default (round 6): switch it off for the fp32 matrix-core form
    soft_f = np.zeros(256)
    eng.reproject(poses, soft=soft_f, tau=TAU, beta=BETA)
    eng.set_option("k2_exact_auto", 1)
    eng.set_option("k2_flags", EXACT)
    tie_f = 0.25 * SCALE * max(abs((soft_f - soft_ref)[i] - (soft_f - soft_ref)[j]) for i, j in pairs)
    print("near-tie weight error: exact %.2e, fast %.2e; max |err - oracle| exact %.2e px" % (tie, tie_f, d.max()))
    assert tie < 0.1 * tie_f
    # the constructed tie of tests/test_gpu_timed_configs.py at the STATED tolerance
    best = int(np.argmax(soft))
    tp = poses.copy()
    tp[(best + 1) % 256] = poses[best] + np.array([1e-9, -1e-9, 1e-9, 1e-6, 1e-6, -1e-6])
    s2 = eng.softInlierScores(tp, tau=TAU, beta=BETA)
    w, _, _ = eng.softMax(s2, SCALE)
    w_ref = orc.softMax(SCALE * orc.soft_inlier(orc.get_diff_maps(tp, fr["xyz"], uv, H, W, cam), TAU, BETA))
    assert np.sort(w_ref)[-2] > 0.2
    margin("a4", "K2 EXACT form: softmax weights in a TIE of the two best hypotheses, scale 0.1: max |w - oracle|", np.abs(w - w_ref).max(), 1e-4)


def test_bench_shape_all_rows_of_a_frame_and_every_entry_point(exact_engine, orc, synth):
    """16 frames x 256 hypotheses x 640x480 in one launch (the timed shape): ALL 256 rows of two of the frames against the oracle, compared on the device
    against the uploaded oracle images; sampled rows of the others; the other entry points launch the same kernel."""
    import torch
    eng = exact_engine
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
    assert n == 1
    ph, sf, wh = out[0].cpu().numpy(), out[3].cpu().numpy(), out[4].cpu().numpy()
    worst_all, worst_s = 0.0, 0.0
    for f in (0, 11):  # every row of two frames
        ref = orc.get_diff_maps(ph[f * N:(f + 1) * N], frames[f]["xyz"], uv, H, W, cam)
        ref_d = torch.from_numpy(ref).to(dev)
        got = err[f * N:(f + 1) * N]
        dd = (got - ref_d).abs()
        dd[((got - CLAMP).abs() <= 1e-3) | ((ref_d - CLAMP).abs() <= 1e-3)] = 0
        worst_all = max(worst_all, float(dd.max().item()))
        assert int((dd > 1e-3).sum().item()) == 0
        soft_ref = orc.soft_inlier(ref, TAU, BETA)
        worst_s = max(worst_s, np.abs(sf[f * N:(f + 1) * N] - soft_ref).max() / soft_ref.max())
        w_ref = orc.softMax(SCALE * soft_ref)
        margin("a4", "K2 EXACT form at the bench shape: softmax weights of a whole frame, scale 0.1: max |w - oracle|", np.abs(wh[f * N:(f + 1) * N] - w_ref).max(), 1e-4)
        del ref_d, dd
    margin("a3", "K2 EXACT form at the bench shape (16 x 256 x 640x480), ALL rows of two frames: max |err - oracle| px", worst_all, 1e-3)
    margin("north*", "K2 EXACT form at the bench shape: soft-inlier scores of two whole frames, relative to the largest score", worst_s, 2e-7, stated=1e-4)
    rng = np.random.default_rng(0)
    worst = 0.0
    for f in range(1, F, 3):
        rows = f * N + rng.choice(N, 6, replace=False)
        got = err[torch.as_tensor(rows, device=dev)].cpu().numpy()
        ref = orc.get_diff_maps(ph[rows], frames[f]["xyz"], uv, H, W, cam)
        mm = excl_clamp_edge(got, ref, CLAMP)
        worst = max(worst, np.abs(got - ref)[mm].max())
    margin("a3", "K2 EXACT form at the bench shape, sampled rows of the other frames: max |err - oracle| px", worst, 1e-3)
    print("exact K2 at the bench shape: %.1f us per launch" % (ms * 1e3))
    err2 = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    soft2 = torch.zeros(F * N, **f64)
    p2 = (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev))
    eng.processImagesBegin(N, err2, seed=4711, thr=10.0, max_tries=1 << 16, soft=soft2, out=p2)
    eng.synchronize()
    assert torch.equal(p2[0], out[0]) and torch.equal(err2, err) and torch.equal(soft2, out[3])
    err2.zero_()
    torch.cuda.synchronize()
    eng.reproject(out[0], N=F * N, err=err2)
    eng.synchronize()
    assert torch.equal(err2, err)


def test_small_and_odd_shapes_and_far_coordinates(exact_engine, orc, synth):
    """The reference's own size (40 x 40 sub-sampled, int16 coordinates, sampled pixel positions), a ragged hypothesis count, a map whose width is not a
    multiple of 64, and coordinates beyond the split's range (|X| >= 65.5 m: those chunks take the fp32 transform -- still inside 1e-3 there, cells at
    scene depth)."""
    eng = exact_engine
    for (h, w, N, q, sampled) in ((40, 40, 256, True, True), (40, 40, 77, True, True), (96, 72, 130, False, False), (120, 200, 64, False, False)):
        fr = synth.chess_like_frame(h, w, seed=99 + h, quantise_int16=q, grid_uv=not sampled)
        uv = fr["uv"]
        eng.set_frame(fr["xyz"], uv if sampled else None, h, w, fr["cam"])
        poses, _, _ = eng.sample(N, seed=5, thr=10.0, max_tries=1 << 16)
        err, soft = np.zeros((N, h * w), np.float32), np.zeros(N)
        eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
        ref = orc.get_diff_maps(poses, fr["xyz"], uv, h, w, fr["cam"])
        m = excl_clamp_edge(err, ref, CLAMP)
        margin("a3", "K2 EXACT form on small / odd maps (%dx%d, N = %d): max |err - oracle| px" % (w, h, N), np.abs(err - ref)[m].max(), 1e-3)
        sr = orc.soft_inlier(ref, TAU, BETA)
        assert np.abs(soft - sr).max() <= 2e-6 * max(1.0, sr.max())
    # coordinates far outside the split's range in a band of the map
    fr = synth.chess_like_frame(H, W, seed=2305)
    xyz = fr["xyz"].copy()
    xyz[100 * W:101 * W] *= 80.0  # one row of cells at |X| up to ~200 m
    uv = synth.pixel_grid(H, W)
    eng.set_frame(xyz, None, H, W, fr["cam"])
    poses, _, _ = eng.sample(128, seed=6, thr=10.0, max_tries=1 << 16)
    err = np.zeros((128, P), np.float32)
    eng.reproject(poses, err=err)
    ref = orc.get_diff_maps(poses, xyz, uv, H, W, fr["cam"])
    m = excl_clamp_edge(err, ref, CLAMP)
    near = np.ones(P, bool)
    near[100 * W:101 * W] = False
    margin("a3", "K2 EXACT form with a band of far coordinates: the cells in range, max |err - oracle| px", np.abs(err - ref)[:, near][m[:, near]].max(), 1e-3)
    if m[:, ~near].any():
        margin("a3", "K2 EXACT form: the far band itself (fp32 transform at |X| ~ 100 m; its points project far outside the image), max |err - oracle| px",
               np.abs(err - ref)[:, ~near][m[:, ~near]].max(), 0.5)


@pytest.mark.parametrize("variant", [84, 85, 89, 93, 94, 95])
def test_every_exact_kernel_form_on_all_rows(exact_engine, orc, synth, variant):
    """The selectable forms of the exact transform (k2_variant with k2_flags bit 28): tiles <64, 256> / <32, 256> / <64, 64>, the tail as reciprocal + Newton step +
    square root (84, 89, 93) or as ONE transcendental, e = n rsq(n z^2) (85, 94, 95 -- the auto policy's since the end of round 6).  All rows of 64 hypotheses on a
    640x480 frame with cells ON the camera plane of a hypothesis (z == 0: projectPoints' z = 1 rule, reached through the chunk's exact-z path) and a zero pose."""
    eng = exact_engine
    fr = synth.chess_like_frame(H, W, seed=77)
    uv, cam = synth.pixel_grid(H, W), fr["cam"]
    xyz = fr["xyz"].copy()
    eng.set_frame(xyz, None, H, W, cam)
    poses, sets, ok = eng.sample(64, seed=5, thr=10.0, max_tries=1 << 16)
    poses[3] = 0.0  # the zero pose: camera frame = scene frame
    xyz[1000:1064, 2] = 0.0  # ... so these cells have z == 0 for hypothesis 3
    xyz[5000] = 0.0          # and this one is the camera centre itself
    eng.set_frame(xyz, None, H, W, cam)
    ref = orc.get_diff_maps(poses, xyz, uv, H, W, cam)
    try:
        eng.set_option("k2_variant", variant)
        err, soft = np.zeros((64, P), np.float32), np.zeros(64)
        eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
    finally:
        eng.set_option("k2_variant", -1)
    m = excl_clamp_edge(err, ref, CLAMP)
    d = np.abs(err - ref)
    d[~m] = 0
    assert d.max() <= 1e-3, "variant %d: %.3e" % (variant, d.max())
    assert np.array_equal(err[3, 1000:1064] >= CLAMP - 1e-3, ref[3, 1000:1064] >= CLAMP - 1e-3)
    soft_ref = orc.soft_inlier(ref, TAU, BETA)
    assert np.abs(soft - soft_ref).max() <= 2e-7 * max(1.0, soft_ref.max())


@pytest.mark.parametrize("cam", [(700.3, 651.7, 301.5, 255.25), (90.0, 90.0, 320.0, 240.0), (1024.0, 1024.0, 320.0, 240.0), (262.5, 262.5, 160.0, 120.0)])
def test_other_intrinsics(exact_engine, orc, synth, cam):
    """Focal lengths other than 7-Scenes' 525: unequal and fractional, short (exponent 7), the largest the split records take (2^10), and a half-resolution
    camera.  All cells of 64 hypotheses on a 640x480 (320x240) map inside the stated 1e-3 px; scores within 2e-7 relative."""
    eng = exact_engine
    h, w = (240, 320) if cam[2] < 200 else (H, W)
    fr = synth.chess_like_frame(h, w, seed=31, cam=cam, grid_uv=True)
    uv = fr["uv"]
    eng.set_frame(fr["xyz"], None, h, w, cam)
    poses, _, _ = eng.sample(64, seed=9, thr=10.0, max_tries=1 << 16)
    err, soft = np.zeros((64, h * w), np.float32), np.zeros(64)
    eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
    ref = orc.get_diff_maps(poses, fr["xyz"], uv, h, w, cam)
    m = excl_clamp_edge(err, ref, CLAMP)
    d = np.abs(err - ref)
    d[~m] = 0
    margin("a3", "K2 EXACT form, camera (%g, %g, %g, %g): max |err - oracle| px over all cells" % cam, d.max(), 1e-3)
    sr = orc.soft_inlier(ref, TAU, BETA)
    assert np.abs(soft - sr).max() <= 2e-7 * max(1.0, sr.max())

"""Pins the oracle restatement (oracle/dsac_oracle.cpp) against the REAL reference sources, compiled where they lie
under /root/reference/core into oracle/_ref/libdsac_ref.so (oracle/refbuild/: OpenCV and Lua/Torch replaced by
stand-ins, everything else is the reference's own code).  Both sides share oracle/cvlike.h for the OpenCV internals
(Rodrigues, projectPoints, solvePnP), so agreement here is expected to floating-point round-off and says: the
restatement of the reference's OWN functions is faithful, statement by statement, including the quirks.

Skipped only where neither the prebuilt library nor /root/reference exists.
"""
import numpy as np
import pytest

from oracle import reference as ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdsac_ref.so not built and /root/reference absent")

H = W = 40
TIGHT = 1e-11


@pytest.fixture(scope="module")
def fr(synth):
    f = synth.chess_like_frame(H, W, seed=3, quantise_int16=True)
    f["uvi"] = f["uv"].astype(np.int32)
    assert np.all(f["uvi"] == f["uv"]) and np.all(np.round(f["xyz"]) == f["xyz"])  # exact in the reference's int / short types
    ref.lib()
    f["camr"] = ref.cam()
    assert np.allclose(f["camr"], f["cam"])
    return f


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def test_pose_conversions(orc, fr):
    rng = np.random.default_rng(1)
    for _ in range(50):
        cv6 = np.concatenate([rng.normal(size=3) * 0.7, rng.normal(size=3) * 800])
        R1, t1 = ref.cv2our(cv6); R2, t2 = orc.cv2our(cv6)
        assert np.abs(R1 - R2).max() <= TIGHT and np.abs(t1 - t2).max() <= TIGHT
        assert np.abs(ref.our2cv(R1, t1) - orc.our2cv(R1, t1)).max() <= TIGHT
        assert np.abs(ref.rodvec_and_trans(R1, t1) - orc.rodvec_and_trans(R1, t1)).max() <= TIGHT
        assert np.abs(ref.cv_to_jp6(cv6) - orc.cv_to_jp6(cv6)).max() <= TIGHT
    # NaN translation -> zero translation (types.h:204-207), identical on both sides
    bad = np.array([0.1, 0.2, 0.3, np.nan, 1.0, 2.0])
    assert np.array_equal(ref.cv2our(bad)[1], orc.cv2our(bad)[1])


def test_getDiffMap_bit_exact(orc, fr):
    rng = np.random.default_rng(2)
    for k in range(8):
        pose = fr["gt_pose"] + rng.normal(size=6) * np.array([0.05, 0.05, 0.05, 60, 60, 60]) * k
        a = ref.getDiffMap(pose, fr["xyz"], fr["uvi"], H, W)
        b = orc.get_diff_maps(pose[None], fr["xyz"], fr["uv"], H, W, fr["camr"])[0].reshape(H, W)
        assert np.array_equal(a, b)  # float32 output of the same double arithmetic
    # rectangular map: the column-major collection order of cnn_softam.h:333-346 must not leak into the layout
    Hr, Wr = 30, 40
    a = ref.getDiffMap(fr["gt_pose"], fr["xyz"][:Hr * Wr], fr["uvi"][:Hr * Wr], Hr, Wr)
    b = orc.get_diff_maps(fr["gt_pose"][None], fr["xyz"][:Hr * Wr], fr["uv"][:Hr * Wr], Hr, Wr, fr["camr"])[0].reshape(Hr, Wr)
    assert np.array_equal(a, b)


def test_project_and_its_jacobians(orc, fr):
    rng = np.random.default_rng(3)
    R, t = orc.cv2our(fr["gt_pose"])
    n_zero = 0
    for k in range(300):
        p = rng.integers(0, H * W)
        pt, obj = fr["uv"][p], fr["xyz"][p].copy()
        if k % 50 == 0:  # behind-the-camera / far-off points: clamp and early-out branches (cnn_softam.h:416,427,476,487)
            obj = obj + rng.normal(size=3).astype(np.float32) * 5000
        assert ref.project(pt, obj, R, t) == orc.project(pt, obj, R, t, fr["camr"])
        Jo = ref.dProjectdObj(pt, obj, R, t); Jh = ref.dProjectdHyp(pt, obj, R, t)
        n_zero += int(not Jo.any())
        assert np.abs(Jo - orc.dProjectdObj(pt, obj, R, t, fr["camr"])).max() <= TIGHT * max(1, np.abs(Jo).max())
        assert np.abs(Jh - orc.dProjectdHyp(pt, obj, R, t, fr["camr"])).max() <= TIGHT * max(1, np.abs(Jh).max())
    assert 0 < n_zero < 300  # both branches were exercised
    # E.z == 0 exactly -> zeros
    obj0 = (np.linalg.inv(R) @ (np.array([10.0, 5.0, 0.0]) - t)).astype(np.float32)
    assert np.array_equal(ref.dProjectdObj(fr["uv"][0], obj0, R, t), orc.dProjectdObj(fr["uv"][0], obj0, R, t, fr["camr"]))


def test_softmax_entropy(orc, fr):
    rng = np.random.default_rng(4)
    for n in (1, 2, 64, 256):
        s = rng.normal(size=n) * 30
        a, b = ref.softMax(s), orc.softMax(s)
        assert np.array_equal(a, b)
        assert ref.entropy(a) == orc.entropy(b)
    s = np.array([1e4, -1e4, 0.0])  # underflow to exact zeros: the dist[i] > 0 guard of entropy (cnn_softam.h:84)
    assert np.array_equal(ref.softMax(s), orc.softMax(s)) and ref.entropy(ref.softMax(s)) == 0.0


def test_p3p_wrapper_and_dPNP(orc, fr):
    rng = np.random.default_rng(5)
    nz = 0
    for _ in range(60):
        idx = rng.choice(H * W, 4, replace=False)
        ok, p_ref = ref.solve_p3p(fr["xyz"][idx], fr["uv"][idx])
        ok2, p_orc = orc.solve_p3p(fr["xyz"][idx], fr["uv"][idx], fr["camr"])
        assert ok == ok2 and np.abs(p_ref - p_orc).max() <= TIGHT * max(1, np.abs(p_ref).max())
        J1 = ref.dPNP(fr["uv"][idx], fr["xyz"][idx]); J2 = orc.dPNP(fr["uv"][idx], fr["xyz"][idx], fr["camr"])
        nz += int(J1.any())
        assert np.abs(J1 - J2).max() <= 1e-9 * max(1, np.abs(J1).max())
    assert nz > 30
    # degenerate set (collinear image points): safeSolvePnP zeroes the pose (cnn_softam.h:66-71); dPNP -> NaN guard -> zeros
    idx = np.array([0, 1, 2, 3])
    X = fr["xyz"][idx].copy(); X[:] = X[0]
    assert np.array_equal(ref.dPNP(fr["uv"][idx], X), orc.dPNP(fr["uv"][idx], X, fr["camr"]))


def test_loss_and_its_gradient(orc, fr):
    rng = np.random.default_rng(6)
    Rg, tg = orc.cv2our(fr["gt_pose"])
    g6 = orc.cv_to_jp6(fr["gt_pose"])
    branches = set()
    for k in range(60):
        if k % 2:  # near the origin the camera centre hardly moves under a rotation: rotation error dominates
            gt = np.concatenate([rng.normal(size=3) * 0.4, rng.normal(size=3) * 5])
            est = gt + rng.normal(size=6) * np.array([0.3, 0.3, 0.3, 1, 1, 1])
        else:
            gt = fr["gt_pose"]
            est = gt + rng.normal(size=6) * np.array([0.001, 0.001, 0.001, 300, 300, 300])
        Rg, tg = orc.cv2our(gt); g6 = orc.cv_to_jp6(gt)
        Re, te = orc.cv2our(est)
        assert abs(ref.maxLoss(Rg, tg, Re, te) - orc.maxLoss(Rg, tg, Re, te)) <= TIGHT * 100
        e6 = orc.cv_to_jp6(est)
        J1, J2 = ref.dLossMax(e6, g6), orc.dLossMax(e6, g6)
        branches.add(bool(J1[3:].any()))
        assert np.abs(J1 - J2).max() <= 1e-9 * max(1, np.abs(J1).max())
    assert branches == {True, False}  # translation-dominated and rotation-dominated cases
    assert not ref.dLossMax(g6, g6).any() and not orc.dLossMax(g6, g6).any()  # zero error -> zero gradient (maxloss.h:133)


def test_refine_and_its_finite_differences(orc, synth, fr):
    perm = synth.refine_permutations(H * W, 8)
    rng = np.random.default_rng(7)
    for k in range(4):
        init = fr["gt_pose"] + rng.normal(size=6) * np.array([0.02, 0.02, 0.02, 25, 25, 25]) * (1 + 3 * k)
        r_ref = ref.refine(init, perm, fr["xyz"], fr["uvi"], H, W)
        out, imap, sd = orc.refine(init[None], perm, fr["xyz"], fr["uv"], H, W, fr["camr"], want_inlier_map=True)
        assert np.abs(orc.cv_to_jp6(out[0]) - r_ref).max() <= 1e-9 * max(1, np.abs(r_ref).max())
        J1 = ref.dRefineHyp(init, perm, fr["xyz"], fr["uvi"], H, W)
        J2 = orc.dRefineHyp(init, perm, fr["xyz"], fr["uv"], H, W, fr["camr"])
        assert np.abs(J1 - J2).max() <= 1e-9 * max(1e-6, np.abs(J1).max())
        if imap.any():
            Jo1 = ref.dRefineObj(init, perm, imap, fr["xyz"], fr["uvi"], H, W, sub_sample=0.05)
            Jo2 = orc.dRefineObj(init, perm, imap, fr["xyz"], fr["uv"], H, W, fr["camr"], sub_sample=0.05)
            assert Jo1.any() or k > 0  # far-off starts may stop at once (< 50 inliers) and give an all-zero Jacobian
            assert np.abs(Jo1 - Jo2).max() <= 1e-9 * max(1e-6, np.abs(Jo1).max())
    # too few inliers: the loop breaks at once and the initial pose comes back (cnn_softam.h:700-701)
    far = fr["gt_pose"] + np.array([1.0, 1.0, 1.0, 3000, 3000, 3000])
    out, sd = orc.refine(far[None], perm, fr["xyz"], fr["uv"], H, W, fr["camr"])
    assert np.abs(orc.cv_to_jp6(out[0]) - ref.refine(far, perm, fr["xyz"], fr["uvi"], H, W)).max() <= 1e-9 * 3000 and sd[0] == 0


def test_dScore_with_both_index_quirks(orc, fr):
    """dScore (cnn_softam.h:564-646): the score script's gradient image is read back transposed (lua_calls.h:329-335)
    and each pixel's 1x3 block lands at column x*40*3 + y*3 (cnn_softam.h:628).  The restatement reproduces both when
    asked to (quirk_transpose + a transposed dDiff) and the plain layout otherwise.  quirk_rot_writeback = the rotation
    matrix re-derived and written back through `const cv::Mat& rot` at every pixel (cnn_softam.h:506-508)."""
    rng = np.random.default_rng(8)
    N = 6
    sets = np.stack([rng.choice(H * W, 4, replace=False) for _ in range(N)]).astype(np.int32)
    pts = np.stack([sets % W, sets // W], -1).astype(np.int32)  # (x, y)
    natural = rng.normal(size=(N, H, W)) * 1e-2  # what the Lua side returns, flattened [hyp][row][col]
    jac = ref.dScore(pts, fr["xyz"], fr["uvi"], ddiff=natural.reshape(N, -1))
    as_read = np.ascontiguousarray(natural.transpose(0, 2, 1)).reshape(N, -1)  # gradients[c](y, x) = table[c][x][y]
    grad, G6, S = orc.dScore(sets, as_read, fr["xyz"], fr["uv"], H, W, fr["camr"], quirk_transpose=True, quirk_rot_writeback=True)
    want = jac.sum(0).reshape(H * W, 3)
    assert np.abs(want).max() > 0
    assert np.abs(grad - want).max() <= 1e-9 * np.abs(want).max()
    # un-quirked restatement = the same numbers at the transposed pixel
    grad_plain, _, _ = orc.dScore(sets, as_read, fr["xyz"], fr["uv"], H, W, fr["camr"], quirk_transpose=False, quirk_rot_writeback=True)
    assert np.abs(grad_plain.reshape(H, W, 3).transpose(1, 0, 2).reshape(-1, 3) - want).max() <= 1e-9 * np.abs(want).max()
    # analytic soft-inlier backward through the same door
    ref.set_score_model(10.0, 0.5, 0.1)
    g = rng.normal(size=N)
    jac2 = ref.dScore(pts, fr["xyz"], fr["uvi"], g=g).sum(0).reshape(H * W, 3)
    # dScore re-solves P3P per set and keeps the pose whether or not the set would have passed the sampling check
    poses = np.stack([orc.solve_p3p(fr["xyz"][s4], fr["uv"][s4], fr["camr"])[1] for s4 in sets])
    err = orc.get_diff_maps(poses, fr["xyz"], fr["uv"], H, W, fr["camr"]).astype(np.float64).reshape(N, H, W)
    s = 1 / (1 + np.exp(-0.5 * (10.0 - err)))
    nat2 = g[:, None, None] * 0.1 * (-0.5) * s * (1 - s)
    grad2, _, _ = orc.dScore(sets, np.ascontiguousarray(nat2.transpose(0, 2, 1)).reshape(N, -1), fr["xyz"], fr["uv"], H, W, fr["camr"], quirk_transpose=True, quirk_rot_writeback=True)
    assert np.abs(grad2 - jac2).max() <= 1e-9 * np.abs(jac2).max()


def test_processImage_forward_and_training_backward(orc, synth, fr):
    """The reference's processImage and the backward section of its training loop on one synthetic frame, replayed
    with the restatement from the reference's own draws (sampling grid, minimal sets, shuffles)."""
    tau, beta, alpha = 10.0, 0.5, 0.1
    ref.set_score_model(tau, beta, alpha)
    N = 32
    gt_jp6 = orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
    r = ref.processImage(1305, fr["xyz"], gt_jp6, hyps=N, backward=True, sub_sample=0.05)
    xyz, uvi = r["estObj"], r["sampling"]
    uv = uvi.astype(np.float32)
    assert np.array_equal(xyz, fr["xyz"])  # metres -> mm -> short survived the stand-in coordinate CNN
    assert uvi[:, 0].min() >= 21 and uvi[:, 0].max() <= 619 and uvi[:, 1].min() >= 21 and uvi[:, 1].max() <= 459
    sets = (r["sampledPoints"][:, :, 1] * W + r["sampledPoints"][:, :, 0]).astype(np.int32)
    # pixelIdxs: one std::shuffle of 0..1599 per step (cnn_softam.h:1104-1114); the order is the C++ library's business
    assert all(np.array_equal(np.sort(row), np.arange(H * W)) for row in r["pixelIdxs"])

    poses, _, ok, _ = orc.sample(N, 0, xyz, uv, H, W, fr["camr"], sets=sets)
    assert ok.all()
    assert np.abs(poses - r["hyps"]).max() <= 1e-9 * np.abs(r["hyps"]).max()
    err = orc.get_diff_maps(poses, xyz, uv, H, W, fr["camr"])
    w = orc.softMax(alpha * orc.soft_inlier(err, tau, beta))
    assert np.abs(w - r["sfScores"]).max() <= 1e-9
    assert abs(orc.entropy(w) - r["sfEntropy"]) <= 1e-9
    avg = orc.avg_pose(w, poses)
    assert np.abs(avg - r["avgHyp"]).max() <= 1e-9 * np.abs(avg).max()
    out, imap, sd = orc.refine(avg[None], r["pixelIdxs"], xyz, uv, H, W, fr["camr"], want_inlier_map=True)
    assert np.abs(out[0] - r["refAvgHyp"]).max() <= 1e-9 * np.abs(out[0]).max()
    assert np.array_equal(imap, r["inlierMap"])
    Re, te = orc.cv2our(out[0]); Rg = orc.rodrigues_vec2mat(gt_jp6[:3]); tg = gt_jp6[3:]
    assert abs(orc.maxLoss(Rg, tg, Re, te) - r["loss"]) <= 1e-9 * max(1, r["loss"])
    rot, tr = orc.pose_errors(Rg, tg, Re, te)
    assert abs(rot - r["rotErr"]) <= 1e-9 and abs(tr - r["tErr"]) <= 1e-9 * max(1, r["tErr"])
    assert r["correct"] == (rot < 5 and tr < 50)

    # backward section (train_ransac_softam.cpp:288-394) from restated pieces, with the reference's index quirks
    dL = orc.dLossMax(orc.cv_to_jp6(out[0]), gt_jp6)
    Jo = orc.dRefineObj(avg, r["pixelIdxs"], imap, xyz, uv, H, W, fr["camr"], sub_sample=0.05)
    Jh = orc.dRefineHyp(avg, r["pixelIdxs"], xyz, uv, H, W, fr["camr"])
    grad = (dL @ Jo).reshape(H * W, 3)
    grad, g = orc.path1_pnp_and_softmax_bwd(dL @ Jh, w, poses, sets, xyz, uv, H, W, fr["camr"], grad=grad)
    s = 1 / (1 + np.exp(-beta * (tau - err.astype(np.float64).reshape(N, H, W))))
    natural = g[:, None, None] * alpha * (-beta) * s * (1 - s)
    grad, _, _ = orc.dScore(sets, np.ascontiguousarray(natural.transpose(0, 2, 1)).reshape(N, -1), xyz, uv, H, W, fr["camr"], quirk_transpose=True, grad=grad,
                            quirk_rot_writeback=True)
    want = r["dLoss_dObj"]
    assert np.abs(want).max() > 0
    assert np.abs(grad - want).max() <= 1e-8 * np.abs(want).max()

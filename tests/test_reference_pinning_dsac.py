"""Pins the restatement of the DSAC (probabilistic selection) variant against the REAL core/cnn.h, compiled where it lies into
oracle/_ref/libdsac_ref_dsac.so (stand-ins for OpenCV / Lua as in tests/test_reference_pinning.py)."""
import numpy as np
import pytest

from oracle import reference_dsac as refd

pytestmark = pytest.mark.skipif(not refd.available(), reason="oracle/_ref/libdsac_ref_dsac.so not built and /root/reference absent")
H = W = 40


@pytest.fixture(scope="module")
def frame(synth, orc):
    fr = synth.chess_like_frame(H, W, seed=11, quantise_int16=True)
    refd.lib()
    tau, beta, alpha = 10.0, 0.5, 0.1
    refd.set_score_model(tau, beta, alpha)
    gt = orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
    r = refd.processImage(1305, fr["xyz"], gt, hyps=24, backward=True, sub_sample=0.05)
    r.update(uv=r["sampling"].astype(np.float32), sets=(r["sampledPoints"][:, :, 1] * W + r["sampledPoints"][:, :, 0]).astype(np.int32), gt=gt,
             cam=np.array(fr["cam"], dtype=np.float64), tau=tau, beta=beta, alpha=alpha)
    return r


def test_forward_all_hypotheses_refined(orc, frame):
    r = frame
    xyz, uv, cam, N = r["estObj"], r["uv"], r["cam"], 24
    poses, _, ok, _ = orc.sample(N, 0, xyz, uv, H, W, cam, sets=r["sets"])
    assert ok.all() and np.abs(poses - r["hyps"]).max() <= 1e-9 * np.abs(r["hyps"]).max()
    err = orc.get_diff_maps(poses, xyz, uv, H, W, cam)
    w = orc.softMax(r["alpha"] * orc.soft_inlier(err, r["tau"], r["beta"]))
    assert np.abs(w - r["sfScores"]).max() <= 1e-12 and abs(orc.entropy(w) - r["sfEntropy"]) <= 1e-12
    assert r["hypIdx"] == int(np.argmax(w))  # randomDraw = false: the most probable hypothesis (cnn.h:122-126)
    Rg = orc.rodrigues_vec2mat(r["gt"][:3])
    for h in range(N):
        ref, imap, sd = orc.refine(r["hyps"][h][None], r["pixelIdxs"], xyz, uv, H, W, cam, want_inlier_map=True)
        imap[r["sets"][h]] = 0  # cnn.h:1208-1214
        assert np.abs(ref[0] - r["refHyps"][h]).max() <= 1e-9 * np.abs(ref[0]).max()
        assert np.array_equal(imap, r["inlierMaps"][h])
        Re, te = orc.cv2our(ref[0])
        assert abs(orc.maxLoss(Rg, r["gt"][3:], Re, te) - r["losses"][h]) <= 1e-9 * max(1, r["losses"][h])
    assert abs(np.dot(w, r["losses"]) - r["expectedLoss"]) <= 1e-9 * max(1, r["expectedLoss"])


def test_refine_from_set_and_dRefine(orc, frame):
    r = frame
    xyz, uv, uvi, cam = r["estObj"], r["uv"], r["sampling"], r["cam"]
    for h in np.argsort(-r["sfScores"])[:3]:
        a = refd.refine_from_set(r["sets"][h], r["pixelIdxs"], xyz, uvi, H, W)
        b = orc.cv_to_jp6(orc.refine_from_set(r["sets"][h], r["pixelIdxs"], xyz, uv, H, W, cam))
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
        J1 = refd.dRefine(r["sets"][h], r["pixelIdxs"], r["inlierMaps"][h], xyz, uvi, H, W, sub_sample=0.05)
        J2 = orc.dRefineDSAC(r["sets"][h], r["pixelIdxs"], r["inlierMaps"][h], xyz, uv, H, W, cam, sub_sample=0.05)
        assert (np.abs(J1).sum(0) > 0).sum() >= 9
        assert np.abs(J1 - J2).max() <= 1e-9 * np.abs(J1).max()


def test_draw_semantics(orc, frame):
    from dsac_amd.engine import Engine
    p = frame["sfScores"]
    assert refd.draw(1, p, random_draw=False) == Engine.draw(p)
    # with the reference's generator the drawn index follows the distribution; the engine's draw is the same map for a given u
    idx = [refd.draw(s, p, random_draw=True) for s in range(200)]
    top = int(np.argmax(p))
    assert abs(np.mean(np.array(idx) == top) - p[top]) < 0.15
    cum = np.cumsum(p[p >= 1e-8])
    for u in (0.0, 0.25, 0.5, 0.999):
        assert Engine.draw(p, u) == int(np.flatnonzero(p >= 1e-8)[np.searchsorted(cum, u * cum[-1], side="right")])
    refd.lib(random_draw=False)


def test_training_backward(orc, frame):
    r = frame
    xyz, uv, cam, N = r["estObj"], r["uv"], r["cam"], 24
    w, sets = r["sfScores"], r["sets"]
    grad = np.zeros((H * W, 3))
    for h in range(N):
        if not w[h] > 1e-4:
            continue
        dL = orc.dLossMax(orc.cv_to_jp6(r["refHyps"][h]), r["gt"])
        grad += w[h] * (dL @ orc.dRefineDSAC(sets[h], r["pixelIdxs"], r["inlierMaps"][h], xyz, uv, H, W, cam, sub_sample=0.05)).reshape(H * W, 3)
    g = w * (r["losses"] - np.dot(w, r["losses"]))
    err = orc.get_diff_maps(r["hyps"], xyz, uv, H, W, cam).astype(np.float64).reshape(N, H, W)
    s = 1 / (1 + np.exp(-r["beta"] * (r["tau"] - err)))
    natural = g[:, None, None] * r["alpha"] * (-r["beta"]) * s * (1 - s)
    # the score script's gradient image is read back transposed (lua_calls.h:329-335); dSMScore then re-orders dScore's
    # column-major blocks to row-major (cnn.h:749-765), so only the first quirk survives
    as_read = np.ascontiguousarray(natural.transpose(0, 2, 1)).reshape(N, -1)
    grad, _, _ = orc.dScore(sets, as_read, xyz, uv, H, W, cam, quirk_transpose=False, grad=grad, quirk_rot_writeback=True)
    want = r["dLoss_dObj"]
    assert np.abs(want).max() > 0
    assert np.abs(grad - want).max() <= 1e-8 * np.abs(want).max()

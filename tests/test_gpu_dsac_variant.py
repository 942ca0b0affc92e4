"""GPU parity of the DSAC-variant rows (core/cnn.h, core/train_ransac.cpp): refinement of all N
hypotheses with per-hypothesis inlier maps, per-hypothesis losses / expected loss, dRefine with minimal-set perturbation,
dSMScore and the trainer's backward section -- HIP engine through the C ABI against the oracle restatement."""
import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu
H = W = 40


@pytest.fixture()
def fwd_state(engine, orc, synth, frame40):
    fr = frame40
    engine.set_frame(fr["xyz"], fr["uv"], H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8)
    gt = orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
    fwd = engine.processImageDSAC(N=48, seed=77, perm=perm, gt_jp6=gt, draw_u=0.37)
    return fr, perm, gt, fwd


def test_refine_all_losses_and_selection(engine, orc, fwd_state):
    fr, perm, gt, fwd = fwd_state
    N = 48
    assert fwd["ok"].all() and abs(fwd["sfScores"].sum() - 1) < 1e-12
    Rg = orc.rodrigues_vec2mat(gt[:3])
    losses = np.zeros(N)
    for h in range(N):
        ref, imap, sd = orc.refine(fwd["hyps"][h][None], perm, fr["xyz"], fr["uv"], H, W, fr["cam"], want_inlier_map=True)
        imap[fwd["sampledPoints"][h]] = 0  # core/cnn.h:1208-1214
        assert sd[0] == fwd["refSteps"][h]
        assert np.array_equal(imap, fwd["inlierMaps"][h])
        assert np.allclose(fwd["refHyps"][h], ref[0], rtol=1e-6, atol=1e-8)
        Re, te = orc.cv2our(ref[0])
        losses[h] = orc.maxLoss(Rg, gt[3:], Re, te)
    assert np.allclose(fwd["losses"], losses, rtol=1e-6, atol=1e-7)
    assert abs(fwd["expectedLoss"] - np.dot(fwd["sfScores"], losses)) <= 1e-6 * max(1, fwd["expectedLoss"])
    # selection: the reference's cumulative-map draw for a given uniform number; arg-max when randomDraw is off
    cum = np.cumsum(fwd["sfScores"])
    assert fwd["hypIdx"] == int(np.searchsorted(cum, 0.37 * cum[-1], side="right"))
    assert engine.draw(fwd["sfScores"]) == int(np.argmax(fwd["sfScores"]))
    # batched dLossMax = per-pose dLossMax
    g = engine.maxLossBatch(fwd["refHyps"], gt, want_grad=True)["grad"]
    for h in (0, 7, 31):
        assert np.allclose(g[h], orc.dLossMax(orc.cv_to_jp6(fwd["refHyps"][h]), gt), rtol=1e-6, atol=1e-9)


def test_drefine_with_minimal_set_perturbation(engine, orc, fwd_state):
    fr, perm, gt, fwd = fwd_state
    order = np.argsort(-fwd["sfScores"])[:3]
    for h in order:
        set4, imap = fwd["sampledPoints"][h], fwd["inlierMaps"][h]
        J_set, px, J_obj = engine.dRefineSet(set4, perm, imap, sub_sample=0.05)
        Jr = orc.dRefineDSAC(set4, perm, imap, fr["xyz"], fr["uv"], H, W, fr["cam"], sub_sample=0.05)
        got = np.zeros_like(Jr)
        for pt in range(3):
            got[:, set4[pt] * 3:set4[pt] * 3 + 3] = J_set[:, pt * 3:pt * 3 + 3]
        for i, p in enumerate(px):
            got[:, p * 3:p * 3 + 3] = J_obj[i]
        assert (np.abs(Jr).sum(0) > 0).sum() >= 9 and len(px) > 0
        scale = np.abs(Jr).max()
        assert np.abs(got - Jr).max() <= 2e-3 * scale, (h, np.abs(got - Jr).max(), scale)
        # the refinement restarted from the set reproduces the forward refinement of that hypothesis
        assert np.allclose(orc.refine_from_set(set4, perm, fr["xyz"], fr["uv"], H, W, fr["cam"]), fwd["refHyps"][h], rtol=1e-6, atol=1e-8)


def test_training_backward_of_the_dsac_variant(engine, orc, fwd_state):
    fr, perm, gt, fwd = fwd_state
    tau, beta = 10.0, 0.5
    bwd = engine.backwardDSAC(fwd, gt, sub_sample=0.05)
    # the same chain from oracle pieces (core/train_ransac.cpp:303-373)
    w, sets = fwd["sfScores"], fwd["sampledPoints"]
    N = len(w)
    grad = np.zeros((H * W, 3))
    for h in range(N):
        if not w[h] > 1e-4:
            continue
        dL = orc.dLossMax(orc.cv_to_jp6(fwd["refHyps"][h]), gt)
        J = orc.dRefineDSAC(sets[h], perm, fwd["inlierMaps"][h], fr["xyz"], fr["uv"], H, W, fr["cam"], sub_sample=0.05)
        grad += w[h] * (dL @ J).reshape(H * W, 3)
    g = w * (fwd["losses"] - np.dot(w, fwd["losses"]))
    assert np.allclose(bwd["scoreOutputGradients"], g, rtol=1e-9, atol=1e-12)
    err = orc.get_diff_maps(fwd["hyps"], fr["xyz"], fr["uv"], H, W, fr["cam"]).astype(np.float64)
    s = 1.0 / (1.0 + np.exp(-beta * (tau - err)))
    dDiff = (g * fwd["score_scale"])[:, None] * (-beta) * s * (1 - s)
    grad, _, _ = orc.dScore(sets, dDiff, fr["xyz"], fr["uv"], H, W, fr["cam"], grad=grad)  # dSMScore re-orders to row-major: no quirk
    emax = np.abs(bwd["grad"] - grad).max() / np.abs(grad).max()
    el2 = np.linalg.norm(bwd["grad"] - grad) / np.linalg.norm(grad)
    print("DSAC-variant end-to-end gradient: max-rel %.3e l2-rel %.3e" % (emax, el2))
    margin("(f)1", "DSAC variant end-to-end gradient vs the oracle's chain: max-rel", emax, 1e-5)
    assert el2 <= 1e-5  # measured 4e-8 vs the reference (tests/test_gpu_reference_golden_dsac.py)


def test_batched_drefine_equals_per_hypothesis_calls(engine, fwd_state):
    """dsac_refine_fd_sets (all weighted hypotheses in one launch of M * (18 + 6 cap) waves) = M calls of dsac_refine_fd_set."""
    fr, perm, gt, fwd = fwd_state
    sel = np.argsort(-fwd["sfScores"])[:6]
    J_set, n_obj, px, J_obj = engine.dRefineSets(fwd["sampledPoints"][sel], perm, fwd["inlierMaps"][sel], sub_sample=0.05)
    assert n_obj.max() > 0
    for i, h in enumerate(sel):
        Js, p1, Jo = engine.dRefineSet(fwd["sampledPoints"][h], perm, fwd["inlierMaps"][h], sub_sample=0.05)
        k = int(n_obj[i])
        assert k == len(p1) and np.array_equal(px[i][:k], p1)
        assert np.array_equal(J_set[i], Js) and np.array_equal(J_obj[i][:k], Jo)
    # empty batch and a batch whose maps hold no inlier cells
    z = engine.dRefineSets(fwd["sampledPoints"][:0], perm, fwd["inlierMaps"][:0])
    assert z[0].shape == (0, 6, 9)
    J0, n0, _, _ = engine.dRefineSets(fwd["sampledPoints"][:2], perm, np.zeros_like(fwd["inlierMaps"][:2]), sub_sample=0.05)
    assert not n0.any() and J0.shape == (2, 6, 9)


def test_selection_expected_loss_and_score_gradients_on_the_device(engine):
    """dsac_select against a literal restatement of the reference: draw (core/cnn.h:102-127: std::map of cumulative sums, upper_bound),
    expectedMaxLoss (:137-150), the score gradients of dSMScore (:737-742)."""
    rng = np.random.default_rng(3)
    for N in (1, 7, 256, 1000):
        w = rng.random(N) ** 6
        w[rng.random(N) < 0.2] = 1e-12  # below EPS: skipped by draw, still part of the expectation
        if N > 3:
            w[3] = 0.0
        w /= w.sum()
        L = rng.random(N) * 50

        def ref_draw(u):
            cum, s, best, bi = {}, 0.0, -1.0, 0
            for i, p in enumerate(w):
                if p < 1e-8:
                    continue
                s += p
                cum[s] = i
                if best < 0 or p > best:
                    best, bi = p, i
            if u is None:
                return bi
            keys = sorted(cum)
            r = u * s
            for k in keys:
                if k > r:
                    return cum[k]
            return cum[keys[-1]]
        for u in [None, 0.0, 0.25, 0.5, 0.999999, float(np.nextafter(1.0, 0.0))] + list(rng.random(5)):
            idx, e, g = engine.selectDSAC(w, L, u)
            assert idx == ref_draw(u), (N, u)
            assert idx == engine.draw(w, u)
        e_ref = 0.0
        for i in range(N):
            e_ref += w[i] * L[i]
        assert e == e_ref
        g_ref = np.array([w[i] * L[i] - sum(w[i] * w[j] * L[j] for j in range(N)) for i in range(N)]) if N <= 256 else w * (L - e_ref)
        assert np.allclose(g, g_ref, rtol=1e-10, atol=1e-15)
        assert np.allclose(g, w * (L - np.dot(w, L)), rtol=1e-9, atol=1e-13)


def test_dsac_variant_replica_plan_on_a_large_map(engine, orc, synth):
    """dsac_refine_fd_sets on a map above 16384 cells: the replica lists come from the tiled two-launch plan (one grid row per hypothesis) instead of
    one workgroup per hypothesis; selected cells = every skip-th inlier of each hypothesis' own map in the reference's column-major order
    (core/cnn.h:935-945), Jacobians = the oracle's dRefine (core/cnn.h:854-990)."""
    H, W = 120, 160
    fr = synth.chess_like_frame(H, W, seed=5)
    engine.set_frame(fr["xyz"], fr["uv"], H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8, seed=3)
    poses, sets, ok, _ = orc.sample(3, 4, fr["xyz"], fr["uv"], H, W, fr["cam"])
    ref, sd, maps = engine.refineAll(poses, perm, sets=sets, want_inlier_maps=True)
    assert (maps > 0).sum(1).min() > 100
    sub = 0.05
    skip = int(1 / sub)
    J_set, n_obj, px, J_obj = engine.dRefineSets(sets, perm, maps, sub_sample=sub)
    for m in range(3):
        order = [y * W + x for x in range(W) for y in range(H) if maps[m][y * W + x] > 0]
        want = order[skip - 1::skip]
        assert [int(p_) for p_ in px[m][:n_obj[m]]] == want
        Jr = orc.dRefineDSAC(sets[m], perm, maps[m], fr["xyz"], fr["uv"], H, W, fr["cam"], sub_sample=sub)
        got = np.zeros_like(Jr)
        for pt in range(3):
            got[:, sets[m][pt] * 3:sets[m][pt] * 3 + 3] = J_set[m][:, pt * 3:pt * 3 + 3]
        for i in range(n_obj[m]):
            p_ = px[m][i]
            got[:, p_ * 3:p_ * 3 + 3] = J_obj[m][i]
        scale = max(np.abs(Jr).max(), 1e-12)
        assert np.abs(got - Jr).max() <= 2e-3 * scale, (m, np.abs(got - Jr).max(), scale)
    # one hypothesis through dsac_refine_fd_set: the same lists
    J1, px1, Jo1 = engine.dRefineSet(sets[1], perm, maps[1], sub_sample=sub)
    assert np.array_equal(px1, px[1][:n_obj[1]]) and np.allclose(J1, J_set[1], rtol=1e-12, atol=1e-15)

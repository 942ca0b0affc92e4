"""HIP engine (through the C ABI) against tests/golden/ref_frame_v1.npz -- outputs of the REAL reference sources
(compiled against OpenCV / Lua stand-ins, see oracle/refbuild/ and tests/golden/make_golden_ref.py) for one
synthetic frame: processImage forward, per-function Jacobians and the training loop's backward section.
The engine replays the frame from the reference's own draws (sampling grid, minimal sets, shuffles).

Tolerances (fp32 projection in K2/K4, fp64 elsewhere; see tests/test_gpu_forward.py for the P3P note):
  error images 1e-3 px (clamp edge excluded) | softmax weights 1e-4 abs | averaged pose 1e-4 rad / 0.05 mm |
  refined pose 1e-5 rel | loss 1e-4 | dScore and end-to-end gradient 1e-4 of the largest entry and in l2 (measured
  1e-6; the reference's own rotation drift, quirk 7, is not reproduced by the product).
"""
import os

import numpy as np
import pytest

from conftest import excl_clamp_edge, margin

pytestmark = pytest.mark.gpu

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = ["ref_frame_v1.npz", "ref_frame_v2.npz"]  # v2: another scene, more noise and outliers, another draw seed (make_golden_ref.py 2)
H = W = 40


@pytest.fixture(scope="module", params=FRAMES)
def g(request):
    d = dict(np.load(os.path.join(GDIR, request.param)))
    d["uv"] = d["sampling"].astype(np.float32)
    d["sets"] = (d["sampledPoints"][:, :, 1] * W + d["sampledPoints"][:, :, 0]).astype(np.int32)
    return d


@pytest.fixture()
def eng(engine, g):
    engine.set_frame(g["estObj"], g["uv"], H, W, tuple(g["cam"]))
    return engine


def pose_delta(a, b):
    from dsac_amd.synth import rodrigues
    D = rodrigues(a[:3]) @ rodrigues(b[:3]).T
    return np.degrees(np.arccos(np.clip((np.trace(D) - 1) / 2, -1, 1))), np.linalg.norm(a[3:] - b[3:]) / max(np.linalg.norm(b[3:]), 1e-9)


def test_hypotheses_from_the_references_minimal_sets(eng, g):
    poses, sets, ok = eng.sample(64, sets=g["sets"], thr=float(g["thr"]))
    assert ok.all() and np.array_equal(sets, g["sets"])  # every set the reference accepted is accepted
    d = np.array([pose_delta(a, b) for a, b in zip(poses, g["hyps"])])
    tight = (d[:, 0] <= 1e-5) & (d[:, 1] <= 1e-6)
    margin("a2", "golden frames (REAL reference): P3P poses from the reference's minimal sets, fraction within 1e-5 deg / 1e-6", tight.mean(), 0.99, at_least=True)
    assert ((d[:, 0] <= 0.1) & (d[:, 1] <= 5e-3)).mean() >= 0.99, (tight.mean(), d.max(0))


def test_error_images_scores_and_soft_argmax(eng, g):
    err = eng.getDiffMap(g["hyps"]).reshape(64, H, W)
    m = excl_clamp_edge(err[:8], g["diffMaps8"])
    margin("a3", "golden frames (REAL reference): error images of 8 hypotheses vs the reference's getDiffMap, max px", np.abs(err[:8] - g["diffMaps8"])[m].max(), 1e-3)
    a = eng.getDiffMap(g["avgHyp"][None]).reshape(H, W)
    assert np.abs(a - g["diffMap_avg"])[excl_clamp_edge(a, g["diffMap_avg"])].max() <= 1e-3
    soft = np.zeros(64)
    eng.reproject(g["hyps"], soft=soft, tau=float(g["tau"]), beta=float(g["beta"]))
    w, ent, avg = eng.softMax(soft, float(g["alpha"]), g["hyps"])
    margin("a4", "golden frames (REAL reference): softmax weights end to end (BASELINE.md 3: 1e-4)", np.abs(w - g["sfScores"]).max(), 1e-4)
    margin("a4", "golden frames (REAL reference): entropy, bits", abs(ent[0] - float(g["sfEntropy"])), 2e-3)
    margin("a5", "golden frames (REAL reference): soft-argmax pose, rotation part (rad)", np.abs(avg[:3] - g["avgHyp"][:3]).max(), 1e-4)
    margin("a5", "golden frames (REAL reference): soft-argmax pose, translation part (mm)", np.abs(avg[3:] - g["avgHyp"][3:]).max(), 5e-2)
    # the fused call (K1 -> K2 -> K3) on the reference's sets gives the same distribution
    out = eng.scoreHypotheses(64, sets=g["sets"], thr=float(g["thr"]), tau=float(g["tau"]), beta=float(g["beta"]), scale=float(g["alpha"]))
    assert np.abs(out[4] - g["sfScores"]).max() <= 1e-4


def test_refinement_and_loss(eng, g):
    got, sd, imap = eng.refine(g["avgHyp"], g["pixelIdxs"], max_inl=int(g["inlier_count"]), thr=float(g["thr"]), want_inlier_map=True)
    assert sd[0] == int(g["ref_steps"]) and np.array_equal(imap, g["inlierMap"])
    margin("a6", "golden frames (REAL reference): refined pose, max |d| / max(1e-2, |component|) (inlier map identical)",
           (np.abs(got[0] - g["refAvgHyp"]) / np.maximum(np.abs(g["refAvgHyp"]), 1e-2)).max(), 1e-5)
    assert np.allclose(got[0], g["refAvgHyp"], rtol=1e-5, atol=1e-7)
    L = eng.maxLoss(g["refAvgHyp"], g["gt_jp6"], want_grad=True)
    margin("a7", "golden frames (REAL reference): maxLoss, relative", abs(L["loss"] - float(g["loss"])) / max(1, float(g["loss"])), 1e-4)
    assert abs(L["rotErr"] - float(g["rotErr"])) <= 1e-4 and abs(L["tErr"] - float(g["tErr"])) <= 1e-3
    assert bool(L["correct"]) == bool(g["correct"])
    margin("a8", "golden frames (REAL reference): dLossMax, max-rel", np.abs(L["grad"] - g["dLossMax"]).max() / np.abs(g["dLossMax"]).max(), 1e-6)


def test_jacobians(eng, g):
    J = eng.dPNP(g["sets"][:8])
    rel = np.array([np.abs(J[h] - g["dPNP8"][h]).max() / max(1.0, np.abs(g["dPNP8"][h]).max()) for h in range(8)])
    margin("a11", "golden frames (REAL reference): dPNP of 8 sets, median max-rel", np.median(rel), 1e-5)
    # 5e-2 until round 5: the P3P solve was built with fused multiply-adds and its replicas could land on another side of an ill-conditioned set's rounding;
    # without contraction the worst of the 8 sets reads 4.4e-6 and the stated tolerance holds for all of them
    margin("a11", "golden frames (REAL reference): dPNP of 8 sets, worst max-rel (SURVEY 8(c): 1e-3)", rel.max(), 1e-3)
    J_hyp, px, J_obj = eng.dRefine(g["avgHyp"], g["pixelIdxs"], g["inlierMap"], max_inl=int(g["inlier_count"]), thr=float(g["thr"]),
                                   sub_sample=float(g["sub_sample"]))
    margin("a14", "golden frames (REAL reference): dRefineHyp, max abs error / (1e-3 max|J| + 1e-6)", np.abs(J_hyp - g["dRefineHyp"]).max() / (1e-6 + 1e-3 * np.abs(g["dRefineHyp"]).max()), 1.0)
    # dScore on an explicit gradient image, read back transposed and written to transposed columns as the reference does
    as_read = np.ascontiguousarray(g["dScore_ddiff_natural"].transpose(0, 2, 1)).reshape(8, -1).astype(np.float32)
    grad = eng.dScore(g["hyps"][:8], g["sets"][:8], as_read, quirk_transpose=True)
    want = g["dScore_jac_sum"]
    emax, el2 = np.abs(grad - want).max() / np.abs(want).max(), np.linalg.norm(grad - want) / np.linalg.norm(want)
    print("dScore vs the reference: max-rel %.3e l2-rel %.3e" % (emax, el2))
    margin("a12", "golden frames (REAL reference): dScore (both index quirks), fp32 K4, max-rel", emax, 1e-5)
    assert el2 <= 1e-5  # measured 1.2e-6
    # the fp64 parity mode with the rotation write-back (quirk 7) on the same input: what is left is the float32 rounding of the d_err volume
    # the C ABI takes (6e-8 relative per entry) -- against the oracle in double on the float32-valued input the mode agrees to 1e-9
    # (tests/test_gpu_backward.py::test_parity_mode_fp64)
    gp = eng.dScore(g["hyps"][:8], g["sets"][:8], as_read, quirk_transpose=True, quirk_rot_writeback=True)
    emax_p = np.abs(gp - want).max() / np.abs(want).max()
    print("dScore, fp64 parity mode with write-back, vs the reference: max-rel %.3e" % emax_p)
    margin("a12", "golden frames (REAL reference): dScore, fp64 parity mode with the rotation write-back, max-rel (float32 d_err input)", emax_p, 1e-6)


def test_training_backward_end_to_end(eng, g):
    tau, beta, alpha = float(g["tau"]), float(g["beta"]), float(g["alpha"])
    fwd = dict(hyps=g["hyps"], sampledPoints=g["sets"], sfScores=g["sfScores"], avgHyp=g["avgHyp"], refAvgHyp=g["refAvgHyp"],
               pixelIdxs=g["pixelIdxs"], inlierMap=g["inlierMap"], refSteps=int(g["ref_steps"]), score_scale=alpha)
    err = eng.getDiffMap(g["hyps"]).reshape(64, H, W).astype(np.float64)

    def d_scores_fn(gs):  # backward of the stand-in score CNN, handed over the way the reference reads it (transposed)
        s = 1 / (1 + np.exp(-beta * (tau - err)))
        natural = gs[:, None, None] * alpha * (-beta) * s * (1 - s)
        return np.ascontiguousarray(natural.transpose(0, 2, 1))

    bwd = eng.backward(fwd, g["gt_jp6"], d_scores_fn=d_scores_fn, thr=float(g["thr"]), inlierCount=int(g["inlier_count"]), tau=tau, beta=beta,
                       sub_sample=float(g["sub_sample"]), quirk_transpose=True)
    want = g["dLoss_dObj"]
    emax = np.abs(bwd["grad"] - want).max() / np.abs(want).max()
    el2 = np.linalg.norm(bwd["grad"] - want) / np.linalg.norm(want)
    print("end-to-end gradient vs the reference: max-rel %.3e l2-rel %.3e" % (emax, el2))
    margin("a15", "golden frames (REAL reference): end-to-end training gradient dLoss/dObj, max-rel", emax, 1e-5)
    assert el2 <= 1e-5  # measured 7e-7

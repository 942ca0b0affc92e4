"""HIP engine (through the C ABI) against tests/golden/refd_frame_v1.npz: outputs of the REAL DSAC-variant reference
(core/cnn.h + the backward section of core/train_ransac.cpp, compiled against OpenCV / Lua stand-ins) for one synthetic
frame, replayed from the reference's own minimal sets and shuffles.  Tolerances as in tests/test_gpu_reference_golden.py."""
import os

import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu
GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = ["refd_frame_v1.npz", "refd_frame_v2.npz"]  # v2: another scene, more noise and outliers, another draw seed (make_golden_ref.py 2)
H = W = 40


@pytest.fixture(scope="module", params=FRAMES)
def g(request):
    d = dict(np.load(os.path.join(GDIR, request.param)))
    d["uv"] = d["sampling"].astype(np.float32)
    d["sets"] = (d["sampledPoints"][:, :, 1] * W + d["sampledPoints"][:, :, 0]).astype(np.int32)
    return d


def test_forward_selection_refinement_expected_loss(engine, g):
    engine.set_frame(g["estObj"], g["uv"], H, W, tuple(g["cam"]))
    N = len(g["sfScores"])
    fwd = engine.processImageDSAC(N=N, sets=g["sets"], perm=g["pixelIdxs"], gt_jp6=g["gt_jp6"], thr=float(g["thr"]), inlierCount=int(g["inlier_count"]),
                                  tau=float(g["tau"]), beta=float(g["beta"]), alpha=float(g["alpha"]), draw_u=None)
    assert fwd["ok"].all()
    assert np.abs(fwd["sfScores"] - g["sfScores"]).max() <= 1e-4 and abs(fwd["sfEntropy"] - float(g["sfEntropy"])) <= 2e-3
    assert fwd["hypIdx"] == int(g["hypIdx"])
    # refinement restarts from the engine's own P3P poses: compare where LM converged from both starts
    close = np.isclose(fwd["refHyps"], g["refHyps"], rtol=1e-5, atol=1e-6).all(1)
    assert close.mean() >= 0.9, close.mean()
    same_maps = np.array([np.array_equal(a, b) for a, b in zip(fwd["inlierMaps"], g["inlierMaps"])])
    assert same_maps.mean() >= 0.9
    assert np.abs(fwd["losses"] - g["losses"])[close].max() <= 1e-4 * max(1.0, np.abs(g["losses"]).max())
    assert abs(fwd["expectedLoss"] - float(g["expectedLoss"])) <= 1e-3 * max(1.0, float(g["expectedLoss"]))
    assert abs(fwd["rotErr"] - float(g["rotErr"])) <= 1e-3 and abs(fwd["tErr"] - float(g["tErr"])) <= 1e-2 and fwd["correct"] == bool(g["correct"])


def test_drefine_and_training_backward(engine, g):
    engine.set_frame(g["estObj"], g["uv"], H, W, tuple(g["cam"]))
    best = int(np.argmax(g["sfScores"]))
    J_set, px, J_obj = engine.dRefineSet(g["sets"][best], g["pixelIdxs"], g["inlierMaps"][best], max_inl=int(g["inlier_count"]), thr=float(g["thr"]),
                                         sub_sample=float(g["sub_sample"]))
    got = np.zeros_like(g["dRefine_best"])
    for pt in range(3):
        got[:, g["sets"][best][pt] * 3:g["sets"][best][pt] * 3 + 3] = J_set[:, pt * 3:pt * 3 + 3]
    for i, p in enumerate(px):
        got[:, p * 3:p * 3 + 3] = J_obj[i]
    assert np.abs(got - g["dRefine_best"]).max() <= 2e-3 * np.abs(g["dRefine_best"]).max()
    # backward section from the reference's forward state
    tau, beta, alpha = float(g["tau"]), float(g["beta"]), float(g["alpha"])
    N = len(g["sfScores"])
    fwd = dict(hyps=g["hyps"], sampledPoints=g["sets"], sfScores=g["sfScores"], refHyps=g["refHyps"], losses=g["losses"], inlierMaps=g["inlierMaps"],
               pixelIdxs=g["pixelIdxs"], score_scale=alpha)
    err = engine.getDiffMap(g["hyps"]).reshape(N, H, W).astype(np.float64)

    def d_scores_fn(gs):  # the stand-in score CNN's backward, handed over as the reference reads it (lua_calls.h:329-335: transposed)
        s = 1 / (1 + np.exp(-beta * (tau - err)))
        return np.ascontiguousarray((gs[:, None, None] * alpha * (-beta) * s * (1 - s)).transpose(0, 2, 1))

    bwd = engine.backwardDSAC(fwd, g["gt_jp6"], d_scores_fn=d_scores_fn, thr=float(g["thr"]), inlierCount=int(g["inlier_count"]),
                              sub_sample=float(g["sub_sample"]))
    want = g["dLoss_dObj"]
    emax = np.abs(bwd["grad"] - want).max() / np.abs(want).max()
    el2 = np.linalg.norm(bwd["grad"] - want) / np.linalg.norm(want)
    print("DSAC-variant end-to-end gradient vs the reference: max-rel %.3e l2-rel %.3e" % (emax, el2))
    margin("(f)1", "DSAC variant, golden frames (REAL cnn.h): end-to-end gradient, max-rel", emax, 1e-3)
    assert el2 <= 1e-3

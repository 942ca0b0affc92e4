"""End-to-end parity: processImage forward + the trainer's backward section, GPU engine vs the same chain
assembled from oracle functions (train_ransac_softam.cpp:288-394).  Tolerance: the chain multiplies several
finite-difference Jacobians; the final gradient must agree to 1 % of its largest entry and 1 % in l2."""
import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu


def oracle_backward(orc, fr, fwd, gt_jp6, tau=10.0, beta=0.5, sub_sample=0.01):
    H, W = fr["H"], fr["W"]
    xyz, uv, cam = fr["xyz"], fr["uv"], fr["cam"]
    poses, sets, w = fwd["hyps"], fwd["sampledPoints"], fwd["sfScores"]
    dL = orc.dLossMax(orc.cv_to_jp6(fwd["refAvgHyp"]), gt_jp6)
    Jo = orc.dRefineObj(fwd["avgHyp"], fwd["pixelIdxs"], fwd["inlierMap"], xyz, uv, H, W, cam, sub_sample=sub_sample)
    Jh = orc.dRefineHyp(fwd["avgHyp"], fwd["pixelIdxs"], xyz, uv, H, W, cam)
    grad = (dL @ Jo).reshape(H * W, 3)
    v6 = dL @ Jh
    grad, g = orc.path1_pnp_and_softmax_bwd(v6, w, poses, sets, xyz, uv, H, W, cam, grad=grad)
    err = orc.get_diff_maps(poses, xyz, uv, H, W, cam).astype(np.float64)
    s = 1.0 / (1.0 + np.exp(-beta * (tau - err)))
    dDiff = (g * fwd["score_scale"])[:, None] * (-beta) * s * (1 - s)
    grad, G6, _ = orc.dScore(sets, dDiff, xyz, uv, H, W, cam, grad=grad)
    return grad, dL, v6, g, w[:, None] * v6[None, :] + G6


def dpnp_substitution(engine, orc, fr, sets, coef6):
    """What the gradient changes by when the oracle's dPNP is replaced by the engine's (K5): the gradient is linear in dPNP,
    grad[cell i of set h] += coef6[h] (1 x 6) . dPNP_h (6 x 12)[:, 3 i .. 3 i + 2], with coef6[h] = w_h v6 (path I) + the pose gradient of
    the score path.  On a near-degenerate minimal set the two dPNP differ by the conditioning of Gao's quartic (test_dpnp_parity), which is
    not what this end-to-end test is about."""
    N = len(sets)
    Jg = np.asarray(engine.dPNP(sets)).reshape(N, 6, 12)
    Jo = np.stack([orc.dPNP(fr["uv"][s_], fr["xyz"][s_], fr["cam"]) for s_ in sets]).reshape(N, 6, 12)
    d = np.einsum("hk,hkc->hc", coef6, Jg - Jo).reshape(N, 4, 3)
    out = np.zeros((fr["H"] * fr["W"], 3))
    np.add.at(out, np.asarray(sets), d)
    return out


def test_process_image_and_backward_reference_size(engine, orc, synth, frame40):
    fr = frame40
    engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    perm = synth.fast_permutations(1600, 8)
    gt_jp6 = orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
    fwd = engine.processImage(N=256, seed=1305, perm=perm, gt_jp6=gt_jp6)
    assert fwd["ok"].all() and fwd["refSteps"] == 8
    assert abs(fwd["sfScores"].sum() - 1) < 1e-12 and 0 < fwd["sfEntropy"] < 8
    Re, te = orc.cv2our(fwd["refAvgHyp"])
    Rg, tg = orc.cv2our(fr["gt_pose"])
    rot, tr = orc.pose_errors(Re, te, Rg, tg)
    assert rot < 1.0 and tr < 20.0  # the synthetic frame is solvable: refined pose within 1 deg / 2 cm of the truth
    # every 5th inlier cell gets its finite-difference replicas (the reference's 1 % leaves 4 cells here, none of which the last
    # refinement steps happen to walk: the gradient would be the round-off of the central differences on both sides)
    bwd = engine.backward(fwd, gt_jp6, sub_sample=0.2)
    ref_grad, dL, v6, g, coef6 = oracle_backward(orc, fr, fwd, gt_jp6, sub_sample=0.2)
    # the oracle's chain as it stands, its own dPNP included: K5 agrees with it to 3e-8 on well-conditioned sets since round 5 (the solve rounds like
    # OpenCV's), so the substitution below -- needed while the two dPNP could differ by 1e-4 ... 5e-2 -- only removes the ill-conditioned sets' share
    margin("a15", "end-to-end training gradient (40x40) vs the oracle's chain WITHOUT substituting the engine's dPNP: max-rel", np.abs(bwd["grad"] - ref_grad).max() / np.abs(ref_grad).max(), 1e-5)
    ref_grad = ref_grad + dpnp_substitution(engine, orc, fr, fwd["sampledPoints"], coef6)
    assert np.abs(bwd["dLoss_dRef"] - dL).max() <= 1e-8 * max(1.0, np.abs(dL).max())
    # v6 = dLoss/dRef . dRefineHyp: when the refinement converges to the same optimum from every start, dRefineHyp is the round-off of its
    # central differences (1e-10) on both sides, hence the absolute floor
    floor = 1e-8 * max(1.0, np.abs(dL).max())
    assert np.abs(bwd["v6"] - v6).max() <= 1e-3 * np.abs(v6).max() + floor
    assert np.abs(bwd["scoreOutputGradients"] - g).max() <= 1e-3 * np.abs(g).max() + floor
    emax = np.abs(bwd["grad"] - ref_grad).max() / np.abs(ref_grad).max()
    el2 = np.linalg.norm(bwd["grad"] - ref_grad) / np.linalg.norm(ref_grad)
    print("end-to-end gradient: max-rel %.3e l2-rel %.3e, nonzero rows %d, max |grad| %.3e (|dLoss/dRef| max %.3e)" %
          (emax, el2, (np.abs(ref_grad).sum(1) > 0).sum(), np.abs(ref_grad).max(), np.abs(dL).max()))
    assert np.abs(ref_grad).max() >= 1e-6 * np.abs(dL).max()  # a real gradient, not the finite differences' round-off
    margin("a15", "end-to-end training gradient dLoss/dObj (40x40, soft-inlier score) vs the oracle's chain: max-rel", emax, 1e-5)
    assert el2 <= 1e-5  # measured 5e-8 (the chain is fp64 except K4's fp32 projection)


def test_process_image_full_resolution_with_score_fn(engine, orc, synth, frame_full):
    """640x480, the score taken from the error images through the score-CNN seam (here: a fixed linear 'CNN')."""
    fr = frame_full
    engine.set_frame(fr["xyz"], None, fr["H"], fr["W"], fr["cam"])
    perm = synth.fast_permutations(fr["H"] * fr["W"], 8)
    gt_jp6 = orc.cv_to_jp6(fr["gt_pose"])

    def score_fn(err):  # higher score for smaller mean error.  The engine hands over a torch DEVICE tensor (the maps stay in HBM), the oracle check an array
        e = err.reshape(err.shape[0], -1)
        return -0.5 * (e.double().mean(dim=1) if hasattr(e, "data_ptr") else e.astype(np.float64).mean(axis=1))

    fwd = engine.processImage(N=64, seed=7, perm=perm, gt_jp6=gt_jp6, score_fn=score_fn)
    ref_err = orc.get_diff_maps(fwd["hyps"], fr["xyz"], fr["uv"], fr["H"], fr["W"], fr["cam"])
    wr = orc.softMax(score_fn(ref_err))
    margin("a4", "processImage 640x480 with a score function on the device-resident error images: softmax weights vs the oracle's from the same poses (BASELINE.md 3)", np.abs(fwd["sfScores"] - wr).max(), 1e-4)
    assert fwd["rotErr"] < 1.0 and fwd["tErr"] < 20.0 and fwd["correct"]

    def d_scores_fn(g):  # backward of the linear 'CNN': d score / d err = -0.5 / P -- returned as a device tensor, which K4 reads in place
        import torch
        P = fr["H"] * fr["W"]
        return torch.as_tensor(g * (-0.5 / P), dtype=torch.float32, device="cuda")[:, None].expand(len(g), P).contiguous()

    bwd = engine.backward(fwd, gt_jp6, d_scores_fn=d_scores_fn)
    assert np.isfinite(bwd["grad"]).all() and np.abs(bwd["grad"]).max() > 0

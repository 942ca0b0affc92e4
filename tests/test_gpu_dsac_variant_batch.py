"""The DSAC variant (core/cnn.h) on FRAME BATCHES (round 5): all N hypotheses of F images refined in ONE launch (dsac_refine_all: F*N waves where one image's
256 leave the chip idle -- SURVEY.md 8(f)1), the per-hypothesis dRefine of the hypotheses that carry weight across images in one launch
(dsac_refine_fd_sets / _frames: core/train_ransac.cpp:314-339), the per-hypothesis losses (dsac_loss_batch_frames: core/cnn.h:137-150) and the selection /
expected loss / dSMScore reductions (dsac_select_frames: core/cnn.h:102-127, 737-742).  Parity: the batch equals F single-frame calls bit for bit; the
single-frame calls are the ones tests/test_gpu_dsac_variant.py and tests/test_gpu_reference_golden_dsac.py pin against the oracle and the real cnn.h."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,F,N,own_uv", [(40, 40, 3, 64, True), (48, 64, 4, 32, False), (480, 640, 2, 16, None),
                                            (37, 53, 2, 24, False), (23, 77, 3, 40, True)])  # the last two: odd sizes, counts no tile divides
def test_dsac_variant_on_a_frame_batch_equals_single_frame_calls(engine, orc, synth, H, W, F, N, own_uv):
    from dsac_amd.capi import lib, ptr, check
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=1200 + f, quantise_int16=(H == 40)) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv = None if own_uv is None else (np.ascontiguousarray(np.stack([fr["uv"] for fr in frames])) if own_uv else frames[0]["uv"])
    uv_of = (lambda f: None) if own_uv is None else ((lambda f: uv[f]) if own_uv else (lambda f: uv))
    cam = frames[0]["cam"]
    perm = synth.fast_permutations(P, 8)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    ctx = engine._ctx
    # hypotheses of every frame: sampled frame by frame (the batch's K1 is pinned elsewhere)
    poses, sets = np.zeros((F * N, 6)), np.zeros((F * N, 4), np.int32)
    for f in range(F):
        engine.set_frame(xyz[f], uv_of(f), H, W, cam)
        p, s, ok = engine.sample(N, seed=60 + f)
        assert ok.all()
        poses[f * N:(f + 1) * N], sets[f * N:(f + 1) * N] = p, s
    # ---- dsac_refine_all: F * N refinement problems in one launch
    engine.set_frames(xyz, uv, H, W, cam, uv_per_frame=bool(own_uv))
    ref_b, sd_b, maps_b = engine.refineAll(poses, perm, sets=sets, want_inlier_maps=True)
    assert (sd_b > 0).any()
    ref_s, sd_s, maps_s = np.zeros_like(ref_b), np.zeros_like(sd_b), np.zeros_like(maps_b)
    for f in range(F):
        engine.set_frame(xyz[f], uv_of(f), H, W, cam)
        sl = slice(f * N, (f + 1) * N)
        ref_s[sl], sd_s[sl], maps_s[sl] = engine.refineAll(poses[sl], perm, sets=sets[sl], want_inlier_maps=True)
    assert np.array_equal(ref_b, ref_s) and np.array_equal(sd_b, sd_s) and np.array_equal(maps_b, maps_s)
    # ---- per-hypothesis losses against each frame's ground truth, selection / expected loss / dSMScore per frame
    out4_b, J_b = np.zeros((F * N, 4)), np.zeros((F * N, 6))
    check(ctx, lib.dsac_loss_batch_frames(ctx, F, N, ptr(ref_b), ptr(gts), ptr(out4_b), ptr(J_b)))
    rng = np.random.default_rng(3)
    w = rng.random((F, N)); w /= w.sum(1, keepdims=True)
    w = np.ascontiguousarray(w.reshape(-1))
    u = np.array([0.3, -1.0, 0.77, 0.999][:F] + [0.5] * max(0, F - 4))
    idx_b, e_b, g_b = np.zeros(F, np.int32), np.zeros(F), np.zeros(F * N)
    check(ctx, lib.dsac_select_frames(ctx, F, N, ptr(w), ptr(out4_b), 4, ptr(u), ptr(idx_b), ptr(e_b), ptr(g_b)))
    for f in range(F):
        sl = slice(f * N, (f + 1) * N)
        L = engine.maxLossBatch(ref_b[sl], gts[f], want_grad=True)
        assert np.array_equal(L["loss"], out4_b[sl, 0]) and np.array_equal(L["grad"], J_b[sl])
        i1, e1, g1 = engine.selectDSAC(w[sl], out4_b[sl, 0], None if u[f] < 0 else float(u[f]))
        assert i1 == idx_b[f] and e1 == e_b[f] and np.array_equal(g1, g_b[sl])
    # ---- dRefine of the hypotheses that carry weight: a different selection per frame, all frames in one launch
    sel = np.concatenate([f * N + np.sort(rng.choice(N, size=2 + f, replace=False)) for f in range(F)]).astype(np.int32)
    frame_of = (sel // N).astype(np.int32)
    M, cap = len(sel), 8
    engine.set_frames(xyz, uv, H, W, cam, uv_per_frame=bool(own_uv))
    Js_b, px_b, Jo_b, n_b = np.zeros((M, 6, 9)), np.zeros((M, cap), np.int32), np.zeros((M, cap, 6, 3)), np.zeros(M, np.int32)
    msel = np.ascontiguousarray(maps_b[sel])
    check(ctx, lib.dsac_refine_fd_sets_frames(ctx, M, ptr(np.ascontiguousarray(sets[sel])), ptr(frame_of), ptr(perm), 8, 100, 50, 10.0, ptr(msel), 0.05, 2.0, ptr(Js_b), ptr(px_b),
                                              ptr(Jo_b), cap, ptr(n_b)))
    assert (n_b > 0).any()
    for f in range(F):
        engine.set_frame(xyz[f], uv_of(f), H, W, cam)
        k = np.flatnonzero(frame_of == f)
        Js, n1, px, Jo = engine.dRefineSets(sets[sel[k]], perm, maps_b[sel[k]], sub_sample=0.05, cap=cap)
        assert np.array_equal(Js, Js_b[k]) and np.array_equal(n1, n_b[k])
        for a, m in enumerate(k):
            assert np.array_equal(px[a][:n1[a]], px_b[m][:n_b[m]]) and np.array_equal(Jo[a][:n1[a]], Jo_b[m][:n_b[m]])
    # the implicit form: M = F x per_frame, hypothesis m in frame m / per_frame
    per = 2
    sel2 = np.concatenate([f * N + np.arange(per) for f in range(F)]).astype(np.int32)
    engine.set_frames(xyz, uv, H, W, cam, uv_per_frame=bool(own_uv))
    Js2, n2, px2, Jo2 = engine.dRefineSets(sets[sel2], perm, maps_b[sel2], sub_sample=0.05, cap=cap)
    Js3, px3, Jo3, n3 = np.zeros_like(Js2), np.zeros_like(px2), np.zeros_like(Jo2), np.zeros_like(n2)
    check(ctx, lib.dsac_refine_fd_sets_frames(ctx, len(sel2), ptr(np.ascontiguousarray(sets[sel2])), ptr((sel2 // N).astype(np.int32)), ptr(perm), 8, 100, 50, 10.0,
                                              ptr(np.ascontiguousarray(maps_b[sel2])), 0.05, 2.0, ptr(Js3), ptr(px3), ptr(Jo3), cap, ptr(n3)))
    assert np.array_equal(Js2, Js3) and np.array_equal(n2, n3)
    with pytest.raises(Exception):
        engine.dRefineSets(sets[sel2[:F * per - 1]], perm, maps_b[sel2[:F * per - 1]], sub_sample=0.05, cap=cap)  # not frames x per_frame
    bad = (sel2 // N).astype(np.int32); bad[0] = F
    with pytest.raises(Exception):
        check(ctx, lib.dsac_refine_fd_sets_frames(ctx, len(sel2), ptr(np.ascontiguousarray(sets[sel2])), ptr(bad), ptr(perm), 8, 100, 50, 10.0,
                                                  ptr(np.ascontiguousarray(maps_b[sel2])), 0.05, 2.0, ptr(Js3), ptr(px3), ptr(Jo3), cap, ptr(n3)))


def test_device_resident_dsac_forward_of_a_frame_batch(engine, orc, synth):
    """Engine.processImagesDSAC (F images per launch chain, everything in HBM) against the per-image, host-orchestrated Engine.processImageDSAC: same
    hypotheses and minimal sets bit for bit, the same refined hypotheses, selection and expected loss up to the fp32 rounding of the soft-inlier scores
    (the batch scores with K1's own staged records, the per-image path re-stages the poses through dsac_reproject)."""
    H, W, F, N = 40, 40, 3, 128
    frames = [synth.chess_like_frame(H, W, seed=1500 + f, quantise_int16=True) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    perm = synth.fast_permutations(H * W, 8)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    u = np.array([0.25, 0.6, 0.9])
    engine.set_frames(xyz, uv, H, W, cam)
    b = engine.processImagesDSAC(N, perm, gts, seed=70, u=u)
    engine.synchronize()
    b = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in b.items()}
    for f in range(F):
        engine.set_frame(xyz[f], uv, H, W, cam)
        s = engine.processImageDSAC(N=N, seed=70 + f, perm=perm, gt_jp6=gts[f], draw_u=float(u[f]))
        sl = slice(f * N, (f + 1) * N)
        assert np.array_equal(b["hyps"][sl], s["hyps"]) and np.array_equal(b["sampledPoints"][sl], s["sampledPoints"])
        assert np.array_equal(b["refHyps"][sl], s["refHyps"]) and np.array_equal(b["refSteps"][sl], s["refSteps"]) and np.array_equal(b["inlierMaps"][sl], s["inlierMaps"])
        assert np.array_equal(b["out4"][sl, 0], s["losses"])
        assert np.abs(b["sfScores"][sl] - s["sfScores"]).max() <= 1e-5
        assert abs(b["expectedLoss"][f] - s["expectedLoss"]) <= 1e-4 * max(1.0, abs(s["expectedLoss"]))
        assert np.abs(b["scoreOutputGradients"][sl] - s["scoreOutputGradients"]).max() <= 1e-4 * max(1.0, np.abs(s["scoreOutputGradients"]).max())

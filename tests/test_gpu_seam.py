"""The score-CNN seam of the BATCHED fast path (round 5): dsac_process_images cut where the reference calls its score CNN.

core/cnn_softam.h:1066-1078 is  getDiffMap x N -> forward(diffMaps) -> softMax ; core/train_ransac_softam.cpp:378-383 is  backward -> dScore.  The pair
dsac_process_images_begin (K1 + K2 of all frames -> error images in HBM) / dsac_process_images_finish (external scores -> K3 -> K6 -> K7) and
dsac_score_backward on the batch carry that seam for F frames per launch chain.

Parity: (a) feeding the soft-inlier scores through finish reproduces dsac_process_images bit for bit, F in {1, 8, 16}, all three deferral modes;
(b) a torch score net through the batched seam (dsac_amd.e2e.ScoredFrameBatch) equals the per-image seam (e2e.TrainStep, tests/test_gpu_e2e.py) frame by
frame and the ORACLE's chain, with both index conventions; (c) the stages one by one on a batch (dsac_sample, dsac_reproject, dsac_softmax_frames)."""
import numpy as np
import pytest

from conftest import excl_clamp_edge, margin

pytestmark = pytest.mark.gpu


def _dev_bufs(torch, dev, F, N, P, maps=True):
    f64 = dict(dtype=torch.float64, device=dev)
    o = dict(hyps=torch.zeros(F * N, 6, **f64), sampledPoints=torch.zeros(F * N, 4, dtype=torch.int32, device=dev), ok=torch.zeros(F * N, dtype=torch.uint8, device=dev),
             scores=torch.zeros(F * N, **f64), sfScores=torch.zeros(F * N, **f64), sfEntropy=torch.zeros(F, **f64), avgHyp=torch.zeros(F, 6, **f64),
             refAvgHyp=torch.zeros(F, 6, **f64), refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, **f64))
    if maps:
        o["inlierMaps"] = torch.zeros(F, P, dtype=torch.int32, device=dev)
    return o


@pytest.mark.parametrize("H,W,F,N", [(40, 40, 1, 256), (40, 40, 8, 256), (40, 40, 16, 128), (480, 640, 1, 256), (480, 640, 8, 128), (480, 640, 16, 128)])
def test_finish_with_the_soft_inlier_scores_equals_process_images(synth, orc, H, W, F, N):
    """(a) begin(soft = the soft-inlier sums) -> finish(scores = those sums, scale = alpha) == dsac_process_images, every output, bit for bit."""
    import torch
    import dsac_amd
    dev = torch.device("cuda", 0)
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=700 + f, quantise_int16=(H == 40)) for f in range(F)]
    xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
    uv = torch.from_numpy(frames[0]["uv"]).to(dev) if H == 40 else None
    cam = frames[0]["cam"]
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    gts = torch.from_numpy(np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])).to(dev)
    with dsac_amd.Engine(0) as eng:
        if F > 1:
            eng.set_frames(xyz, uv, H, W, cam, borrow=True)
        else:
            eng.set_frame(xyz[0], uv, H, W, cam, borrow=True)
        ref = _dev_bufs(torch, dev, F, N, P)
        err_ref = torch.empty(F * N, P, dtype=torch.float32, device=dev)
        eng.processImages(N, perm, gt_jp6=gts, seed=91, err=err_ref, out=ref)
        eng.synchronize()
        assert bool(ref["ok"].all()) and int(ref["refSteps"].min()) == 8
        for mode in (0, 1, 2):
            eng.set_option("pi_defer_tail", mode)
            reps = 3 if mode else 1  # deferred: consecutive pairs (alternating arrays) must not disturb each other
            outs = [_dev_bufs(torch, dev, F, N, P) for _ in range(2)]
            errs = [torch.empty(F * N, P, dtype=torch.float32, device=dev) for _ in range(2)]
            for i in range(reps):
                o, e = outs[i & 1], errs[i & 1]
                eng.processImagesBegin(N, e, seed=91, soft=o["scores"], out=(o["hyps"], o["sampledPoints"], o["ok"]))
                eng.processImagesFinish(N, o["scores"], perm, o["hyps"], gt_jp6=gts, scale=0.1, out=o)
            eng.joinTail()
            eng.synchronize()
            for i in range(min(reps, 2)):
                o, e = outs[i], errs[i]
                for key in ref:
                    assert torch.equal(o[key], ref[key]), (key, mode, i)
                assert torch.equal(e, err_ref), ("err", mode, i)
        eng.set_option("pi_defer_tail", 0)
        # finish without its begin, or with another shape, is refused
        with pytest.raises(Exception):
            eng.processImagesFinish(N, ref["scores"], perm, ref["hyps"], gt_jp6=gts, scale=0.1, out=_dev_bufs(torch, dev, F, N, P))
        # host arrays through the same pair (staged copies, in stream order)
        if H == 40:
            eng.set_frames(xyz.cpu().numpy(), None if uv is None else uv.cpu().numpy(), H, W, cam) if F > 1 else eng.set_frame(xyz[0].cpu().numpy(), uv.cpu().numpy(), H, W, cam)
            e_h, soft_h = np.zeros((F * N, P), np.float32), np.zeros(F * N)
            ph, sh, okh = eng.processImagesBegin(N, e_h, seed=91, soft=soft_h)
            r = eng.processImagesFinish(N, soft_h, perm.cpu().numpy(), ph, gt_jp6=gts.cpu().numpy(), scale=0.1, want_inlier_maps=True)
            assert np.array_equal(ph, ref["hyps"].cpu().numpy()) and np.array_equal(e_h, err_ref.cpu().numpy())
            for key in ("sfScores", "sfEntropy", "avgHyp", "refAvgHyp", "refSteps", "inlierMaps", "out4"):
                assert np.array_equal(r[key], ref[key].cpu().numpy()), key


def test_stages_one_by_one_on_a_frame_batch(engine, synth):
    """(c) dsac_sample / dsac_reproject / dsac_softmax_frames on a batch == dsac_score_hypotheses_frames (K1, K2, K3 of every frame), bit for bit."""
    H, W, F, N = 48, 64, 3, 128
    frames = [synth.chess_like_frame(H, W, seed=820 + f) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    engine.set_frames(xyz, uv, H, W, cam)
    err_ref = np.zeros((F * N, H * W), np.float32)
    poses_r, sets_r, ok_r, sc_r, w_r, ent_r, avg_r = engine.scoreHypothesesFrames(N, seed=5, err=err_ref)
    poses, sets, ok = engine.sample(F * N, seed=5, max_tries=1 << 20)
    assert np.array_equal(poses, poses_r) and np.array_equal(sets, sets_r) and np.array_equal(ok, ok_r)
    err, soft = np.zeros((F * N, H * W), np.float32), np.zeros(F * N)
    engine.reproject(poses, err=err, soft=soft)
    # dsac_reproject re-stages the poses through fp64 Rodrigues (K1 hands K2 its own records): the error images agree to fp32 rounding, like Engine.processImage
    m = excl_clamp_edge(err, err_ref)
    margin("a3", "dsac_reproject on a frame batch vs the fused K1 -> K2 of dsac_score_hypotheses_frames, max px", np.abs(err - err_ref)[m].max(), 1e-3)
    w, ent, avg = engine.softMaxFrames(sc_r, N, scale=0.1, poses=poses_r)
    assert np.array_equal(w, w_r) and np.array_equal(ent, ent_r) and np.array_equal(avg, avg_r)
    # one frame at a time agrees with the batch: frame f scored from the stream of seed + f
    for f in range(F):
        engine.set_frame(xyz[f], uv, H, W, cam)
        p1, s1, o1 = engine.sample(N, seed=5 + f)
        assert np.array_equal(p1, poses[f * N:(f + 1) * N]) and np.array_equal(s1, sets[f * N:(f + 1) * N])
    engine.set_frames(xyz, uv, H, W, cam)
    with pytest.raises(Exception):
        engine.sample(F * N, seed=5, sets=sets)  # given sets are evaluated frame by frame
    with pytest.raises(Exception):
        engine.reproject(poses[:F * 64])  # hypotheses per frame not a multiple of 128


class _Table:
    def __new__(cls, xyz_m):
        import torch

        class _T(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.table = torch.nn.Parameter(torch.as_tensor(xyz_m, dtype=torch.float32))

            def forward(self, patches):
                return self.table + 0.0 * patches.mean()
        return _T()


@pytest.mark.parametrize("quirk", [False, True])
def test_score_net_through_the_batched_seam(synth, orc, quirk):
    """(b) F frames x N hypotheses with the reference's score-CNN architecture at the seam: every frame equals the per-image path (e2e.TrainStep) and frame 0
    the oracle's chain -- error images in (n, y, x) order, softmax of the CNN's scores, refinement, loss, and the scene-coordinate gradient through the
    CNN's own autograd and dScore; quirk = the reference's index conventions (lua_calls.h:329-335 with cnn_softam.h:628,641)."""
    import torch
    from dsac_amd import e2e
    S, F, N, sub = 40, 4, 128, 0.05
    P = S * S
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    net = e2e.ScoreNet().to(dev)
    frames = [synth.chess_like_frame(S, S, seed=900 + f, quantise_int16=True) for f in range(F)]
    perm = synth.fast_permutations(P, 8)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    # what CNN 1 would hand over: metres -> mm in float32, exactly as TrainStep forms it
    xyz_d = torch.stack([(torch.as_tensor(fr["xyz"] / 1000.0, dtype=torch.float32, device=dev) * 1000.0).float() for fr in frames]).contiguous()
    uv_d = torch.stack([torch.as_tensor(fr["uv"], device=dev) for fr in frames]).contiguous()
    sb = e2e.ScoredFrameBatch(0, frames=F, hyps=N, sub_sample=sub, score_net=net)
    perm_d, gt_d = torch.as_tensor(perm, device=dev), torch.as_tensor(gts, device=dev)
    res = sb.forward(xyz_d, uv_d, gt_d, perm_d, seed=1305)
    for p in net.parameters():
        p.grad = None
    grad_b = sb.backward(quirk_transpose=quirk).clone()
    torch.cuda.synchronize()
    net_grads_b = [p.grad.clone() for p in net.parameters()]
    assert bool(sb.ok.all()) and int(res["refSteps"].min()) == 8
    # ---- every frame against the per-image path
    patches = torch.rand(P, 3, 42, 42, device=dev) * 255
    net_sum = [torch.zeros_like(g) for g in net_grads_b]
    for f in range(F):
        ts = e2e.TrainStep(0, hyps=N, sub_sample=sub, coord_net=_Table(frames[f]["xyz"] / 1000.0), score_net=net)
        for p in net.parameters():
            p.grad = None
        out = ts.forward_backward(patches, uv_d[f], gts[f], perm, seed=1305 + f, quirk_transpose=quirk)
        torch.cuda.synchronize()
        hs = slice(f * N, (f + 1) * N)
        assert torch.equal(sb.sets[hs], ts.sets) and torch.equal(sb.poses[hs], ts.poses)
        # TrainStep runs dsac_reproject (poses re-staged through fp64 Rodrigues), the batch K1's own records: fp32 rounding of the error images
        eb, e1 = sb.err[hs].cpu().numpy().reshape(N, P), ts.err.cpu().numpy().reshape(N, P)
        m = excl_clamp_edge(eb, e1)
        margin("(f)2", "batched seam vs per-image seam: error images the score CNN reads, max px", np.abs(eb - e1)[m].max(), 1e-3)
        margin("(f)2", "batched seam vs per-image seam: softmax weights of the CNN's scores", float((res["sfScores"][hs] - ts.w).abs().max()), 1e-4)
        margin("(f)2", "batched seam vs per-image seam: refined pose, max-rel", float((res["refAvgHyp"][f] - ts.ref).abs().max() / max(1.0, float(ts.ref.abs().max()))), 1e-5)
        g1 = ts.grad_xyz
        margin("(f)2", "batched seam vs per-image seam: scene-coordinate gradient, max / max|g|", float((grad_b[f] - g1).abs().max() / g1.abs().max()), 1e-3)
        for a, p in zip(net_sum, net.parameters()):
            a += p.grad
        ts.engine.close()
    # the score CNN's parameter gradients of the batch = the sum over the per-image passes (one backward over F*N maps)
    for a, b in zip(net_sum, net_grads_b):
        assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-30)
    # ---- frame 0 against the ORACLE's chain (as tests/test_gpu_e2e.py::test_seam_against_the_oracle does for the per-image path)
    f = 0
    cam, xyz, uvh = sb.cam, xyz_d[f].cpu().numpy(), frames[f]["uv"]
    hs = slice(0, N)
    poses, sets = sb.poses[hs].cpu().numpy(), sb.sets[hs].cpu().numpy()
    err_o = orc.get_diff_maps(poses, xyz, uvh, S, S, cam)
    err_g = sb.err[hs].cpu().numpy().reshape(N, P)
    m = excl_clamp_edge(err_g, err_o, 100.0)
    margin("(f)2", "batched seam: K2's tensor vs the oracle's getDiffMap of every hypothesis (n, y, x order), max px", np.abs(err_g - err_o)[m].max(), 1e-3)
    scores = sb.scores[hs].cpu().numpy()
    w_o = orc.softMax(scores)
    assert np.abs(res["sfScores"][hs].cpu().numpy() - w_o).max() <= 1e-12
    avg_o = orc.avg_pose(w_o, poses)
    assert np.abs(avg_o - res["avgHyp"][f].cpu().numpy()).max() <= 1e-9 * max(1.0, np.abs(avg_o).max())
    ref_o, imap_o, _ = orc.refine(avg_o, perm, xyz, uvh, S, S, cam, want_inlier_map=True)
    margin("(f)2", "batched seam: refined pose vs the oracle's chain from the CNN's scores, max-rel",
           np.abs(ref_o[0] - res["refAvgHyp"][f].cpu().numpy()).max() / max(1.0, np.abs(ref_o).max()), 1e-6)
    assert np.array_equal(imap_o, res["inlierMaps"][f].cpu().numpy())
    dL = orc.dLossMax(orc.cv_to_jp6(ref_o[0]), gts[f])
    Jh = orc.dRefineHyp(avg_o, perm, xyz, uvh, S, S, cam)
    Jo = orc.dRefineObj(avg_o, perm, imap_o, xyz, uvh, S, S, cam, sub_sample=sub)
    grad_o, g_o = orc.path1_pnp_and_softmax_bwd(dL @ Jh, w_o, poses, sets, xyz, uvh, S, S, cam, grad=(dL @ Jo).reshape(P, 3))
    e = torch.as_tensor(err_g.reshape(N, 1, S, S), device=dev).requires_grad_(True)
    net(e).backward(gradient=torch.as_tensor(g_o, device=dev).float().clamp_(-0.1, 0.1))
    G = e.grad.reshape(N, S, S).double().cpu().numpy()
    dDiff = G.transpose(0, 2, 1) if quirk else G
    grad_o, _, _ = orc.dScore(sets, dDiff, xyz, uvh, S, S, cam, quirk_transpose=quirk, grad=grad_o)
    got = grad_b[f].cpu().numpy()
    scale = np.abs(grad_o).max()
    assert scale > 0
    p3p_o = np.stack([orc.solve_p3p(xyz[s_], uvh[s_], cam)[1] for s_ in sets])
    same = np.abs(p3p_o - poses).max(1) <= 1e-6 * np.maximum(1.0, np.abs(poses).max(1))
    assert same.all(), "K1 poses that differ from the oracle's P3P of the same sets: %s" % np.flatnonzero(~same)  # all of them agree since round 5 (csrc/dmath.h)
    rel = np.abs(got - grad_o).max(1) / scale
    margin("(f)2", "batched seam: scene-coordinate gradient through the CNN's own autograd vs the oracle's chain, max / max|g|", rel.max(), 1e-5)
    sb.engine.close()


def test_process_image_with_a_device_score_function_keeps_the_maps_in_hbm(engine, synth, orc, frame_full):
    """Engine.processImage(score_fn): the function receives a torch DEVICE tensor that aliases what K2 wrote (round 4 shipped 314 MB per image through NumPy)."""
    import torch
    fr = frame_full
    engine.set_frame(fr["xyz"], None, fr["H"], fr["W"], fr["cam"])
    perm = synth.fast_permutations(fr["H"] * fr["W"], 8)
    seen = {}

    def score_fn(err):
        seen["cuda"], seen["shape"], seen["ptr"] = err.is_cuda, tuple(err.shape), err.data_ptr()
        return -0.5 * err.reshape(err.shape[0], -1).double().mean(dim=1)
    fwd = engine.processImage(N=64, seed=7, perm=perm, gt_jp6=orc.cv_to_jp6(fr["gt_pose"]), score_fn=score_fn, keep_err=True)
    assert seen["cuda"] and seen["shape"] == (64, fr["H"], fr["W"]) and isinstance(fwd["diffMaps"], torch.Tensor) and fwd["diffMaps"].data_ptr() == seen["ptr"]
    assert fwd["correct"] and fwd["refSteps"] == 8


@pytest.mark.parametrize("H,W,F,N", [(40, 40, 1, 64), (48, 64, 3, 128)])
def test_soft_inlier_gradient_images_reproduce_the_in_kernel_form(engine, synth, H, W, F, N):
    """dsac_soft_score_derr (the soft-inlier score's backward written out as the gradient images a score model hands to dScore) + dsac_score_backward ==
    dsac_soft_score_backward (the same derivative formed inside K4), to fp32 rounding; and the images equal the formula evaluated in numpy."""
    from dsac_amd.capi import lib, ptr, check
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=40 + f, quantise_int16=(H == 40)) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    if F > 1:
        engine.set_frames(xyz, uv, H, W, cam)
    else:
        engine.set_frame(xyz[0], uv, H, W, cam)
    err, soft = np.zeros((F * N, P), np.float32), np.zeros(F * N)
    poses, sets, ok = engine.processImagesBegin(N, err, seed=3, soft=soft)
    g = np.random.default_rng(1).standard_normal(F * N) * 1e-2
    d_err = np.zeros((F * N, P), np.float32)
    check(engine._ctx, lib.dsac_soft_score_derr(engine._ctx, F * N, ptr(g), ptr(err), 100.0, 10.0, 0.5, ptr(d_err)))
    e = err.astype(np.float64)
    s = 1.0 / (1.0 + np.exp(-0.5 * (10.0 - e)))
    want = np.where(e >= 100.0, 0.0, g[:, None] * (-0.5) * s * (1.0 - s))
    margin("(f)2", "dsac_soft_score_derr vs the formula in numpy: max |d| / max |d_err|", np.abs(d_err - want).max() / np.abs(want).max(), 1e-5)
    J = engine.dPNP(sets)
    a = engine.dScore(poses, sets, d_err, dpnp=J)
    b = engine.dSoftScore(poses, sets, g, tau=10.0, beta=0.5, dpnp=J)
    margin("(f)2", "explicit gradient images through dsac_score_backward vs the in-kernel soft-score backward: max |d| / max |g|", np.abs(a - b).max() / np.abs(b).max(), 1e-4)


def test_upload_right_after_a_deferred_call_waits_for_the_tail(synth, orc):
    """dsac_copy_async / dsac_fill_zero_async into memory a deferred tail still reads (the borrowed frame, the ground truth) are ordered behind
    that tail -- the refinement of call i must see frame i, not the frame uploaded for call i + 1."""
    import torch
    import dsac_amd
    from dsac_amd.capi import lib, ptr, check
    dev = torch.device("cuda", 0)
    H, W, N = 480, 640, 256
    P = H * W
    fa, fb = synth.chess_like_frame(H, W, seed=11), synth.chess_like_frame(H, W, seed=12)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    gts = [torch.from_numpy(orc.cv_to_jp6(f["gt_pose"])).to(dev).view(1, 6) for f in (fa, fb)]
    xa, xb = torch.from_numpy(fa["xyz"]).to(dev), torch.from_numpy(fb["xyz"]).to(dev)
    with dsac_amd.Engine(0) as eng:
        want = []
        for x, gt in ((xa, gts[0]), (xb, gts[1])):  # in order, each frame on its own
            eng.set_frame(x, None, H, W, fa["cam"], borrow=True)
            o = _dev_bufs(torch, dev, 1, N, P)
            eng.processImages(N, perm, gt_jp6=gt, seed=5, out=o)
            eng.synchronize()
            want.append({k: v.clone() for k, v in o.items()})
        buf = xa.clone()      # ONE frame buffer, refilled between the calls
        gt_buf = gts[0].clone()
        outs = [_dev_bufs(torch, dev, 1, N, P) for _ in range(2)]
        for mode in (1, 2):
            eng.set_option("pi_defer_tail", mode)
            for rep in range(3):
                check(eng._ctx, lib.dsac_copy_async(eng._ctx, ptr(buf), ptr(xa), P * 12))
                check(eng._ctx, lib.dsac_copy_async(eng._ctx, ptr(gt_buf), ptr(gts[0]), 48))
                eng.set_frame(buf, None, H, W, fa["cam"], borrow=True)
                eng.processImages(N, perm, gt_jp6=gt_buf, seed=5, out=outs[0])
                # the tail of that call (K6 on one wave: ~100 us) is still refining against `buf` when the next frame is uploaded into it
                check(eng._ctx, lib.dsac_copy_async(eng._ctx, ptr(buf), ptr(xb), P * 12))
                check(eng._ctx, lib.dsac_copy_async(eng._ctx, ptr(gt_buf), ptr(gts[1]), 48))
                eng.set_frame(buf, None, H, W, fa["cam"], borrow=True)
                eng.processImages(N, perm, gt_jp6=gt_buf, seed=5, out=outs[1])
                eng.joinTail()
                eng.synchronize()
                for k in want[0]:
                    assert torch.equal(outs[0][k], want[0][k]) and torch.equal(outs[1][k], want[1][k]), (k, mode, rep)
        eng.set_option("pi_defer_tail", 0)


def test_score_hypotheses_frames_with_the_score_tail_deferred(synth):
    """dsac_score_hypotheses_frames under "pi_defer_tail" = 2 (the bench's default step since round 5): the reduction of the per-tile sums and K3 of call i
    run on the tail stream beside K1 of call i + 1; alternating result arrays, one error-image buffer.  Every call's results equal the in-order call's."""
    import torch
    import dsac_amd
    dev = torch.device("cuda", 0)
    H, W, F, N = 480, 640, 4, 128
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=30 + f) for f in range(F)]
    xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)

    def bufs():
        f64 = dict(dtype=torch.float64, device=dev)
        return (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev),
                torch.zeros(F * N, **f64), torch.zeros(F * N, **f64), torch.zeros(F, **f64), torch.zeros(F, 6, **f64))
    err = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    with dsac_amd.Engine(0) as eng:
        eng.set_frames(xyz, None, H, W, frames[0]["cam"], borrow=True)
        want = []
        for i in range(4):
            o = bufs()
            eng.scoreHypothesesFrames(N, seed=100 + i, err=err, out=o)
            eng.synchronize()
            want.append([t.clone() for t in o] + [err.clone()] if i == 3 else [t.clone() for t in o])
        eng.set_option("pi_defer_tail", 2)
        sets = [bufs(), bufs()]
        got = []
        for i in range(4):
            eng.scoreHypothesesFrames(N, seed=100 + i, err=err, out=sets[i & 1])
            if i >= 1:  # call i - 1's arrays are complete once call i's launches are ordered behind its tail: join, then copy them out in stream order
                pass
        eng.joinTail()
        eng.synchronize()
        # the last two calls' arrays are intact (calls 2 and 3); the error images are those of call 3
        for i in (2, 3):
            for a, b in zip(sets[i & 1], want[i][:7]):
                assert torch.equal(a, b), i
        assert torch.equal(err, want[3][7])
        # with the SAME arrays every call the contract is broken on purpose nowhere: the in-order mode is what a caller with one set of arrays uses
        eng.set_option("pi_defer_tail", 0)
        o = bufs()
        eng.scoreHypothesesFrames(N, seed=103, err=err, out=o)
        eng.synchronize()
        for a, b in zip(o, want[3][:7]):
            assert torch.equal(a, b)

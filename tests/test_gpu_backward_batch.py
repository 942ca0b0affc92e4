"""The training backward on FRAME BATCHES (round 4): dsac_dpnp, dsac_backward_path1 (dLossMax -> dRefineObj / dRefineHyp -> contraction -> dPNP -> support
scatter + softmax backward), dsac_refine_fd, dsac_soft_score_backward and dsac_score_backward with a batch set by dsac_set_frames -- one launch per stage
for all frames, one P x 3 gradient per frame.  core/train_ransac_softam.cpp:288-394 is one image per round; a batch is what a data-parallel step puts on
one GPU.

Parity: (1) the batch equals F single-frame calls -- bit for bit for the fp64 chain (dLossMax, K5, the finite-difference replicas and their Jacobians,
v6, the score gradients), to 1e-12 where fp64 atomics sum several hypotheses' terms into a shared support cell, and to fp32 rounding (1e-5 of the
largest entry) for the score backward K4, whose fp32 partial sums are grouped by the launch's workgroup count (32 workgroups per frame in a batch
of 16, 512 for a single frame); (2) the batch result of every frame against the ORACLE's chain."""
import numpy as np
import pytest

from conftest import margin
from test_gpu_pipeline import dpnp_substitution, oracle_backward

pytestmark = pytest.mark.gpu


def _grid_frame(synth, H, W, seed, cam):
    """A 'chess-like' frame whose pixel positions ARE the map's grid (u = x, v = y) under a camera scaled to the map -- what the engine's implicit grid
    assumes (synth.chess_like_frame places a small map's cells on a stride-4 sub-sample of a 640 x 480 image)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = cam
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    rvec = axis * np.deg2rad(rng.uniform(0, 30))
    tvec = (rng.uniform(-1, 1, size=3) + np.array([0, 0, 2.5])) * 1000.0
    R = synth.rodrigues(rvec)
    uv = np.stack(np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32)), -1).reshape(-1, 2)
    P = H * W
    depth = rng.uniform(800.0, 3500.0, size=P)
    Xc = np.stack([(uv[:, 0] - cx) / fx * depth, (uv[:, 1] - cy) / fy * depth, depth], -1)
    Xgt = (Xc - tvec) @ R
    xyz = Xgt + rng.normal(scale=20.0, size=(P, 3))
    out = rng.uniform(size=P) < 0.3
    xyz[out] = Xgt.mean(0) + rng.uniform(-2000.0, 2000.0, size=(int(out.sum()), 3))
    return dict(xyz=xyz.astype(np.float32), uv=uv, gt_pose=np.concatenate([rvec, tvec]), H=H, W=W, cam=tuple(float(c) for c in cam))


@pytest.mark.parametrize("H,W,F,N,implicit", [(40, 40, 3, 128, False), (120, 160, 2, 256, True), (40, 40, 3, 128, "own"),
                                              (37, 53, 2, 128, True), (45, 31, 2, 128, False)])  # the last two: odd sizes that no tile divides
def test_backward_on_a_frame_batch(engine, orc, synth, H, W, F, N, implicit):
    """implicit = False: one table of image positions shared by the frames; True: the implicit grid; "own": one table PER FRAME (uv_per_frame -- the
    reference's sub-sampled maps, whose positions stochasticSubSample draws per image, core/cnn_softam.h:283-309)."""
    own = implicit == "own"
    implicit = implicit is True
    P = H * W
    frames = ([_grid_frame(synth, H, W, 300 + f, (525.0 * W / 640, 525.0 * W / 640, W / 2.0, H / 2.0)) for f in range(F)] if implicit else
              [synth.chess_like_frame(H, W, seed=300 + f, quantise_int16=(H == 40)) for f in range(F)])
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv = None if implicit else (np.ascontiguousarray(np.stack([fr["uv"] for fr in frames])) if own else frames[0]["uv"])
    uv_of = (lambda f: uv[f]) if own else (lambda f: uv)
    # the oracle's view of the pixel positions: the engine's implicit grid is u = x, v = y of the MAP (synth.pixel_grid is the stride-4 sampling of a 640 x 480 image)
    uvh = np.stack(np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32)), -1).reshape(-1, 2) if implicit else frames[0]["uv"]
    cam = frames[0]["cam"]
    perm = synth.fast_permutations(P, 8)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    alpha, tau, beta, sub = 0.1, 10.0, 0.5, 0.2
    # forward of the batch (dsac_process_images): what the backward starts from
    engine.set_frames(xyz, uv, H, W, cam, uv_per_frame=own)
    fwd = engine.processImages(N, perm, gt_jp6=gts, seed=41, want_inlier_maps=True)
    assert fwd["ok"].all() and (fwd["refSteps"] == 8).all()
    rng = np.random.default_rng(5)
    d_err = (rng.standard_normal((F * N, P)) * 1e-3).astype(np.float32)

    def backward(nf, sl, fsl):
        """path I + softmax backward, then the soft-score backward, then an explicit d_err volume on top; nf frames are set in the engine"""
        J = np.zeros((nf * N, 6, 12))
        r = engine.backwardPath1(fwd["hyps"][sl], fwd["sampledPoints"][sl], fwd["sfScores"][sl], fwd["avgHyp"][fsl], fwd["refAvgHyp"][fsl], gts[fsl], perm,
                                 fwd["inlierMaps"][fsl], sub_sample=sub, out_dpnp=J)
        g_path1 = r["grad"].copy()
        g_soft = engine.dSoftScore(fwd["hyps"][sl], fwd["sampledPoints"][sl], r["g"] * alpha, tau=tau, beta=beta, dpnp=J)
        G6_soft = engine.lastPoseGradients(nf * N)
        g_derr = engine.dScore(fwd["hyps"][sl], fwd["sampledPoints"][sl], d_err[sl], dpnp=J)
        Jh, px, Jo, n = engine.dRefineFrames(fwd["avgHyp"][fsl], perm, fwd["inlierMaps"][fsl], sub_sample=sub, cap=64)
        return dict(path1=g_path1, g=r["g"].copy(), dL=np.asarray(r["dL"]).reshape(nf, 6), v6=np.asarray(r["v6"]).reshape(nf, 6), dpnp=J, soft=g_soft, G6=G6_soft,
                    derr=g_derr, Jh=Jh, px=px, Jo=Jo, n=n, dpnp_call=engine.dPNP(fwd["sampledPoints"][sl]))

    b = backward(F, slice(0, F * N), slice(0, F))
    assert b["path1"].shape == (F * P, 3) and (b["n"] > 0).all()
    for f in range(F):
        engine.set_frame(xyz[f], uv_of(f), H, W, cam)
        if own:  # the forward of the batch as well: frame f with ITS table, bit for bit
            s1 = engine.processImages(N, perm, gt_jp6=gts[f:f + 1], seed=41 + f, want_inlier_maps=True)
            for key in ("hyps", "sampledPoints", "sfScores"):
                assert np.array_equal(fwd[key][f * N:(f + 1) * N], s1[key]), key
            assert np.array_equal(fwd["refAvgHyp"][f], s1["refAvgHyp"][0]) and np.array_equal(fwd["inlierMaps"][f], s1["inlierMaps"][0])
        s = backward(1, slice(f * N, (f + 1) * N), slice(f, f + 1))
        hs, ps = slice(f * N, (f + 1) * N), slice(f * P, (f + 1) * P)
        # the fp64 chain: bit for bit
        for key, sl_ in (("g", hs), ("dL", slice(f, f + 1)), ("v6", slice(f, f + 1)), ("dpnp", hs), ("dpnp_call", hs), ("Jh", slice(f, f + 1)),
                         ("px", slice(f, f + 1)), ("Jo", slice(f, f + 1)), ("n", slice(f, f + 1))):
            assert np.array_equal(b[key][sl_], s[key]), (key, f)
        # ... except where fp64 ATOMICS add several hypotheses' support terms into one cell (train_ransac_softam.cpp:344-358: minimal sets share cells):
        # the order of three or more additions is the hardware's, in either form -- two runs of the same call differ by as much
        margin("a15", "frame batch vs single-frame call, path-I gradient (fp64 atomics on shared support cells): max |d| / max |g|",
               np.abs(b["path1"][ps] - s["path1"]).max() / max(np.abs(s["path1"]).max(), 1e-300), 1e-12)
        # K4 (fp32 partial sums grouped by the launch's workgroup count): to fp32 rounding
        for key in ("soft", "derr"):
            margin("a15", "frame batch vs single-frame call, K4 gradient (%s): max |d| / max |g|" % key, np.abs(b[key][ps] - s[key]).max() / max(np.abs(s[key]).max(), 1e-300), 1e-5)
        margin("a10", "frame batch vs single-frame call, K4 pose sums: max-rel", (np.abs(b["G6"][hs] - s["G6"]).max(1) / np.maximum(np.abs(s["G6"]).max(1), 1e-9 * np.abs(s["G6"]).max() + 1e-300)).max(), 1e-4)
        # ... and the batch's frame f against the oracle's chain (core/train_ransac_softam.cpp:288-394 for that image)
        if own:
            uvh = uv[f]
        fr = dict(frames[f], uv=uvh)
        fw = dict(hyps=fwd["hyps"][hs], sampledPoints=fwd["sampledPoints"][hs], sfScores=fwd["sfScores"][hs], avgHyp=fwd["avgHyp"][f], refAvgHyp=fwd["refAvgHyp"][f],
                  pixelIdxs=perm, inlierMap=fwd["inlierMaps"][f], score_scale=alpha)
        ref_grad, dL, v6, g, coef6 = oracle_backward(orc, fr, fw, gts[f], tau=tau, beta=beta, sub_sample=sub)
        ref_grad = ref_grad + dpnp_substitution_frame(engine, orc, fr, xyz[f], uv_of(f), H, W, cam, fw["sampledPoints"], coef6)
        got = b["soft"][ps]  # path I + softmax backward + soft-score backward of frame f (dSoftScore accumulated onto a fresh gradient: add path I)
        got = got + b["path1"][ps]
        scale = np.abs(ref_grad).max()
        assert scale >= 1e-6 * np.abs(dL).max()
        # the oracle re-solves P3P from the sets: every pose agrees with K1's since round 5 (OpenCV's arithmetic and alignment, csrc/dmath.h); until then
        # the support cells of the 1-2 % of ill-conditioned sets had to be left out of this comparison
        p3p_o = np.stack([orc.solve_p3p(fr["xyz"][s_], uvh[s_], cam)[1] for s_ in fw["sampledPoints"]])
        same = np.abs(p3p_o - fw["hyps"]).max(1) <= 1e-6 * np.maximum(1.0, np.abs(fw["hyps"]).max(1))
        assert same.all(), "K1 poses that differ from the oracle's P3P of the same sets: %s" % np.flatnonzero(~same)
        margin("a15", "frame batch: end-to-end gradient of every frame vs the oracle's chain, max-rel", np.abs(got - ref_grad).max() / scale, 1e-5)
        margin("a8", "frame batch: dLossMax of every frame vs oracle", np.abs(b["dL"][f] - dL).max() / max(1.0, np.abs(dL).max()), 1e-8)
    engine.set_frames(xyz, uv, H, W, cam, uv_per_frame=own)
    with pytest.raises(Exception):
        engine.dSoftScore(fwd["hyps"][:F * N - 7], fwd["sampledPoints"][:F * N - 7], np.zeros(F * N - 7))  # not frames x (hypotheses per frame)


def dpnp_substitution_frame(engine, orc, fr, xyz_f, uv, H, W, cam, sets, coef6):
    """tests/test_gpu_pipeline.dpnp_substitution with the engine's K5 evaluated on THIS frame (the engine holds the batch otherwise)"""
    import dsac_amd
    with dsac_amd.Engine(0) as e2:
        e2.set_frame(xyz_f, uv, H, W, cam)
        return dpnp_substitution(e2, orc, fr, sets, coef6)


def test_score_backward_into_managed_memory(engine, synth):
    """The main pass of K4 adds into grad_xyz with hardware fp64 atomics, which are only defined on ordinary device memory: a MANAGED gradient buffer
    (hipMallocManaged) takes the staged form by itself -- on one frame and, frame by frame, on a batch -- and receives the same gradient (to the fp32
    grouping of the partial sums) as a torch device tensor does."""
    import ctypes
    import torch
    from dsac_amd.capi import lib, ptr, check
    hip = ctypes.CDLL("libamdhip64.so")
    H, W, F, N = 40, 40, 2, 128
    P = H * W
    dev = torch.device("cuda", 0)
    frames = [synth.chess_like_frame(H, W, seed=610 + f, quantise_int16=True) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    rng = np.random.default_rng(11)
    d_err = torch.from_numpy((rng.standard_normal((F * N, P)) * 1e-3).astype(np.float32)).to(dev)
    managed = ctypes.c_void_p()
    nbytes = F * P * 3 * 8
    assert hip.hipMallocManaged(ctypes.byref(managed), ctypes.c_size_t(nbytes), ctypes.c_uint(1)) == 0
    try:
        view = np.ctypeslib.as_array(ctypes.cast(managed, ctypes.POINTER(ctypes.c_double)), shape=(F * P, 3))
        for nf in (1, F):
            if nf == 1:
                engine.set_frame(xyz[0], uv, H, W, cam)
            else:
                engine.set_frames(xyz, uv, H, W, cam)
            poses, sets, ok = engine.sample(nf * N, seed=3)
            poses_d, sets_d = torch.from_numpy(poses).to(dev), torch.from_numpy(sets).to(dev)
            J = torch.from_numpy(np.asarray(engine.dPNP(sets))).to(dev)
            ref = torch.zeros(nf * P, 3, dtype=torch.float64, device=dev)
            check(engine._ctx, lib.dsac_score_backward(engine._ctx, nf * N, ptr(poses_d), ptr(sets_d), ptr(d_err[:nf * N]), ptr(J), 0, ptr(ref)))
            engine.synchronize()
            view[:] = 0.0
            check(engine._ctx, lib.dsac_score_backward(engine._ctx, nf * N, ptr(poses_d), ptr(sets_d), ptr(d_err[:nf * N]), ptr(J), 0, managed.value))
            engine.synchronize()
            torch.cuda.synchronize()
            r = ref.cpu().numpy()
            assert np.abs(r).max() > 0
            margin("a9", "K4 into a managed gradient buffer (staged form; %d frame(s)) vs into device memory: max |d| / max |g|" % nf,
                   np.abs(view[:nf * P] - r).max() / np.abs(r).max(), 1e-5)
    finally:
        hip.hipFree(managed)


@pytest.mark.parametrize("Nf", [120, 72])
def test_score_backward_on_a_batch_with_any_hypothesis_count_and_in_parity_mode(engine, orc, synth, Nf):
    """One launch for all frames needs 16 | hypotheses per frame (several tiles per frame beyond 256: the test below).  Other counts (120, 72: not multiples of 16) and the fp64
    parity mode run frame by frame inside the call: the batch's gradient and pose sums equal the single-frame calls' -- bit for bit in parity mode
    (a sequential fp64 recurrence; the single-frame mode is pinned against the oracle in tests/test_gpu_backward.py), to the last bit of the fp64 atomics
    otherwise."""
    H, W, F = 40, 40, 3
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=770 + f, quantise_int16=True) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    rng = np.random.default_rng(Nf)
    d_err = (rng.standard_normal((F * Nf, P)) * 1e-3).astype(np.float32)
    engine.set_frames(xyz, uv, H, W, cam)
    poses, sets, ok = engine.sample(F * Nf, seed=9) if Nf % 128 == 0 else (None, None, None)
    if poses is None:  # dsac_sample on a batch draws whole 128-hypothesis groups; any count: frame by frame
        ps, ss = [], []
        for f in range(F):
            engine.set_frame(xyz[f], uv, H, W, cam)
            p1, s1, o1 = engine.sample(Nf, seed=9 + f)
            ps.append(p1); ss.append(s1)
        poses, sets = np.concatenate(ps), np.concatenate(ss)
        engine.set_frames(xyz, uv, H, W, cam)
    J = np.asarray(engine.dPNP(sets))
    g_b = engine.dScore(poses, sets, d_err, dpnp=J)
    G6_b = engine.lastPoseGradients(F * Nf)
    g_p = engine.dScore(poses, sets, d_err, dpnp=J, parity_fp64=True) if Nf <= 128 else None
    for f in range(F):
        hs, cs = slice(f * Nf, (f + 1) * Nf), slice(f * P, (f + 1) * P)
        engine.set_frame(xyz[f], uv, H, W, cam)
        g1 = engine.dScore(poses[hs], sets[hs], d_err[hs], dpnp=J[hs])
        G6_1 = engine.lastPoseGradients(Nf)
        margin("a15", "frame batch with %d hypotheses per frame (frame by frame inside the call) vs single-frame calls, K4 gradient: max |d| / max |g|" % Nf,
               np.abs(g_b[cs] - g1).max() / np.abs(g1).max(), 1e-12)
        assert np.array_equal(G6_b[hs], G6_1)
        if g_p is not None:
            gp1 = engine.dScore(poses[hs], sets[hs], d_err[hs], dpnp=J[hs], parity_fp64=True)
            assert np.array_equal(g_p[cs], gp1)
        engine.set_frames(xyz, uv, H, W, cam)


@pytest.mark.parametrize("Nf", [384, 512])
def test_more_than_256_hypotheses_per_frame_in_one_launch(engine, orc, synth, Nf):
    """Round 6 (core/train_ransac_softam.cpp:380-383): a frame batch with MORE than 256 hypotheses per frame runs its score backward as ONE
    main-pass launch -- several equal hypothesis tiles per frame (384 -> 2 x 192, 512 -> 2 x 256), every tile adding into its frame's gradient with fp64
    atomics as the tiles of a single big frame do -- instead of frame by frame.  Against the single-frame calls: the fp32 partial sums are grouped by other
    tiles (a single frame of 384 runs as 256 + 128), so the agreement is fp32 rounding, as for the other batch shapes."""
    H, W, F = 40, 40, 3
    P = H * W
    frames = [synth.chess_like_frame(H, W, seed=880 + f, quantise_int16=True) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    rng = np.random.default_rng(Nf)
    d_err = (rng.standard_normal((F * Nf, P)) * 1e-3).astype(np.float32)
    engine.set_frames(xyz, uv, H, W, cam)
    poses, sets, ok = engine.sample(F * Nf, seed=9)
    J = np.asarray(engine.dPNP(sets))
    engine.profile_enable(True, stride=1)
    engine.profile_read(1, reset=True)
    g_b = engine.dScore(poses, sets, d_err, dpnp=J)
    ms, n = engine.profile_read(1, reset=True)
    assert n == 1, "the batch's score backward took %d main-pass launches" % n
    G6_b = engine.lastPoseGradients(F * Nf)
    engine.profile_enable(False)
    for f in range(F):
        hs, cs = slice(f * Nf, (f + 1) * Nf), slice(f * P, (f + 1) * P)
        engine.set_frame(xyz[f], uv, H, W, cam)
        g1 = engine.dScore(poses[hs], sets[hs], d_err[hs], dpnp=J[hs])
        G6_1 = engine.lastPoseGradients(Nf)
        margin("a15", "frame batch with %d hypotheses per frame in ONE launch vs single-frame calls, K4 gradient: max |d| / max |g|" % Nf,
               np.abs(g_b[cs] - g1).max() / np.abs(g1).max(), 1e-5)
        margin("a10", "frame batch with %d hypotheses per frame in ONE launch vs single-frame calls, pose sums G6: max |d| / max |G6|" % Nf,
               np.abs(G6_b[hs] - G6_1).max() / np.abs(G6_1).max(), 1e-5)
        if f == 0:  # ... and frame 0 against the oracle's dScore (the engine's dPNP of the same sets is oracle-tested on its own)
            ref, _, _ = orc.dScore(sets[hs], d_err[hs].astype(np.float64), frames[0]["xyz"], uv, H, W, cam)
            margin("a15", "frame batch with %d hypotheses per frame in ONE launch, frame 0 against the oracle's dScore: max |d| / max |g|" % Nf,
                   np.abs(g_b[cs] - ref).max() / np.abs(ref).max(), 1e-3)
        engine.set_frames(xyz, uv, H, W, cam)

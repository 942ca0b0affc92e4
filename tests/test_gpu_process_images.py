"""dsac_process_images: the reference's whole test-time unit of work -- processImage of core/cnn_softam.h:960-1179 as
core/test_ransac_softam.cpp:97-157 calls it per image -- for a batch of frames, one launch per stage (K1, K2, K3, K6 with one wave per frame,
K7).  Parity: a batch equals F single-frame processImage calls (seeds seed + f) bit for bit, and the single-frame call is the one the oracle /
golden tests pin (tests/test_gpu_pipeline.py, tests/test_gpu_reference_golden.py); the refinement and the loss of every frame are also compared
with the oracle directly."""
import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,F,N", [(40, 40, 5, 128), (48, 64, 3, 256)])
def test_batch_equals_single_frame_process_image(engine, orc, synth, H, W, F, N):
    frames = [synth.chess_like_frame(H, W, seed=500 + f, quantise_int16=(H == 40)) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv = frames[0]["uv"]
    cam = frames[0]["cam"]
    perm = synth.fast_permutations(H * W, 8)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    engine.set_frames(xyz, uv, H, W, cam)
    err = np.zeros((F * N, H * W), np.float32)
    b = engine.processImages(N, perm, gt_jp6=gts, seed=77, err=err, want_inlier_maps=True)
    assert b["ok"].all() and (b["refSteps"] == 8).all()
    for f in range(F):
        engine.set_frame(xyz[f], uv, H, W, cam)
        e1 = np.zeros((N, H * W), np.float32)
        s = engine.processImages(N, perm, gt_jp6=gts[f:f + 1], seed=77 + f, err=e1, want_inlier_maps=True)  # the same export on ONE frame
        sl = slice(f * N, (f + 1) * N)
        assert np.array_equal(b["sampledPoints"][sl], s["sampledPoints"]) and np.array_equal(b["hyps"][sl], s["hyps"])
        assert np.array_equal(err[sl], e1)
        assert np.array_equal(b["sfScores"][sl], s["sfScores"]) and b["sfEntropy"][f] == s["sfEntropy"][0]
        assert np.array_equal(b["avgHyp"][f], s["avgHyp"][0]) and np.array_equal(b["refAvgHyp"][f], s["refAvgHyp"][0])
        assert np.array_equal(b["inlierMaps"][f], s["inlierMaps"][0]) and b["refSteps"][f] == s["refSteps"][0]
        assert np.array_equal(b["out4"][f], s["out4"][0])
        # ... and the host-orchestrated mirror of processImage (Engine.processImage: separate C-ABI calls, poses re-staged through fp64 Rodrigues)
        m = engine.processImage(N=N, seed=77 + f, perm=perm, gt_jp6=gts[f])
        assert np.array_equal(m["sampledPoints"], s["sampledPoints"]) and np.array_equal(m["hyps"], s["hyps"])
        assert np.abs(m["sfScores"] - s["sfScores"]).max() <= 1e-6 and np.abs(m["refAvgHyp"] - s["refAvgHyp"][0]).max() <= 1e-5 * max(1.0, np.abs(m["refAvgHyp"]).max())
        assert abs(m["loss"] - s["out4"][0][0]) <= 1e-5 * max(1.0, m["loss"]) and m["correct"] == bool(s["out4"][0][3])
        # the oracle on the same soft-argmax pose: refinement (core/cnn_softam.h:1099-1154) and loss (core/maxloss.h:69-79)
        ref_o, imap_o, sd_o = orc.refine(b["avgHyp"][f], perm, xyz[f], uv, H, W, cam, want_inlier_map=True)
        margin("a6", "dsac_process_images (frame batch): refined pose of every frame vs oracle, max-rel (inlier maps identical)", np.abs(ref_o[0] - b["refAvgHyp"][f]).max() / max(1.0, np.abs(ref_o).max()), 1e-7)
        assert np.array_equal(imap_o, b["inlierMaps"][f])
        R1, t1 = orc.cv2our(b["refAvgHyp"][f])
        R2 = orc.rodrigues_vec2mat(gts[f][:3])
        loss_o = orc.maxLoss(R1, t1, R2, gts[f][3:])
        margin("a7", "dsac_process_images (frame batch): loss of every frame vs oracle, relative", abs(loss_o - b["out4"][f][0]) / max(1.0, loss_o), 1e-9)
    # the batched refinement / loss exports on their own
    engine.set_frames(xyz, uv, H, W, cam)
    from dsac_amd.capi import lib, ptr, check
    ref2, sd2, maps2 = np.zeros((F, 6)), np.zeros(F, np.int32), np.zeros((F, H * W), np.int32)
    check(engine._ctx, lib.dsac_refine(engine._ctx, F, ptr(np.ascontiguousarray(b["avgHyp"])), ptr(perm), 8, 100, 50, 10.0, None, None, ptr(ref2), ptr(maps2), ptr(sd2)))
    assert np.array_equal(ref2, b["refAvgHyp"]) and np.array_equal(maps2, b["inlierMaps"]) and np.array_equal(sd2, b["refSteps"])
    assert np.array_equal(engine.maxLossFrames(ref2, gts)["out4"], b["out4"])
    with pytest.raises(Exception):
        engine.processImages(100, perm)  # not a multiple of 128 on a frame batch


def test_process_images_at_baseline_size_on_device(engine, synth):
    """640 x 480, 4 frames x 256 hypotheses, every buffer on the GPU: the batch reproduces the single-frame calls bit for bit."""
    import torch
    H, W, F, N = 480, 640, 4, 256
    P = H * W
    dev = torch.device("cuda", 0)
    frames = [synth.chess_like_frame(H, W, seed=1305 + f) for f in range(F)]
    xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    gts = torch.zeros(F, 6, dtype=torch.float64, device=dev)

    def bufs(nf):
        n = nf * N
        return dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev),
                    ok=torch.zeros(n, dtype=torch.uint8, device=dev), scores=torch.zeros(n, dtype=torch.float64, device=dev),
                    sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(nf, dtype=torch.float64, device=dev),
                    avgHyp=torch.zeros(nf, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(nf, 6, dtype=torch.float64, device=dev),
                    refSteps=torch.zeros(nf, dtype=torch.int32, device=dev), out4=torch.zeros(nf, 4, dtype=torch.float64, device=dev))
    b = bufs(F)
    err = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    engine.set_frames(xyz, None, H, W, frames[0]["cam"], borrow=True)
    engine.processImages(N, perm, gt_jp6=gts, seed=9, err=err, out=b)
    engine.synchronize()
    assert int(b["ok"].sum().item()) == F * N and bool((b["refSteps"] == 8).all())
    for f in range(F):
        s = bufs(1)
        e1 = torch.empty(N, P, dtype=torch.float32, device=dev)
        engine.set_frame(xyz[f], None, H, W, frames[0]["cam"], borrow=True)
        engine.processImages(N, perm, gt_jp6=gts[f:f + 1], seed=9 + f, err=e1, out=s)
        engine.synchronize()
        sl = slice(f * N, (f + 1) * N)
        assert torch.equal(b["hyps"][sl], s["hyps"]) and torch.equal(err[sl], e1)
        # a batch of 4 x 256 x 640x480 takes the big-launch K2 form, one frame the single-frame form: same error images, scores summed in another order
        assert torch.allclose(b["sfScores"][sl], s["sfScores"], rtol=1e-6, atol=1e-12)
        assert torch.allclose(b["refAvgHyp"][f], s["refAvgHyp"][0], rtol=1e-6, atol=1e-6) and torch.allclose(b["out4"][f], s["out4"][0], rtol=1e-6, atol=1e-6)
    del err
    torch.cuda.empty_cache()


def test_deferred_refinement_tail_gives_the_same_results(engine, synth):
    """dsac_set_option("pi_defer_tail", 1): K6 / K7 of a batch run on their own stream under K1 / K2 of the next batch (2: the score reduction and
    K3 as well).  Three batches of different
    frames in a row, every tail output in its own buffer, read after joinTail: bit-equal to the same three calls in stream order; a host
    destination switches the deferral off for that call; other entry points see a pending tail's results in stream order."""
    import torch
    H, W, F, N = 120, 160, 4, 128
    P = H * W
    dev = torch.device("cuda", 0)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    cam = None
    batches = []
    for k in range(3):
        frames = [synth.chess_like_frame(H, W, seed=900 + 10 * k + f) for f in range(F)]
        cam = frames[0]["cam"]
        batches.append(torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev))
    gts = torch.zeros(F, 6, dtype=torch.float64, device=dev)

    def bufs():
        n = F * N
        return dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev),
                    ok=torch.zeros(n, dtype=torch.uint8, device=dev), scores=torch.zeros(n, dtype=torch.float64, device=dev),
                    sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(F, dtype=torch.float64, device=dev),
                    avgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev),
                    refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, dtype=torch.float64, device=dev),
                    inlierMaps=torch.zeros(F, P, dtype=torch.int32, device=dev))

    def run(defer):
        engine.set_option("pi_defer_tail", int(defer))
        shared = bufs()          # everything but the tail's outputs is shared by the three calls, as a loop over batches would do
        own = [bufs() for _ in range(3)]
        # torch zero-fills the new buffers on ITS stream, the engine writes them from its own (non-blocking) streams: without this a fill can land after
        # the engine's write when the queues are time-sliced (seen late in a full test session, never in a short one)
        torch.cuda.synchronize(dev)
        outs = []
        for k in range(3):
            o = dict(shared)
            # mode 2 (score tail deferred too): K3 of a call reads its poses / scores beside K1 of the next -- every output in its own buffer
            for key in (own[k] if int(defer) == 2 else ("refAvgHyp", "refSteps", "out4", "inlierMaps", "avgHyp", "sfScores")):
                o[key] = own[k][key]
            engine.set_frames(batches[k], None, H, W, cam, borrow=True)
            engine.processImages(N, perm, gt_jp6=gts, seed=31 + k, out=o)
            outs.append(o)
        engine.joinTail()
        engine.synchronize()
        return [{key: v.cpu().numpy().copy() for key, v in o.items()} for o in outs]

    try:
        plain = run(False)
        deferred = run(True)
        both_tails = run(2)
        for a, b, c2 in zip(plain, deferred, both_tails):
            assert (a["refSteps"] == 8).all() and a["ok"].all()
            for key in a:
                assert np.array_equal(a[key], b[key]), key
            # shared buffers of the in-order run hold the LAST call's values: compare what every call owns in both
            for key in ("refAvgHyp", "refSteps", "out4", "inlierMaps", "avgHyp", "sfScores"):
                assert np.array_equal(a[key], c2[key]), key
        for key in plain[2]:
            assert np.array_equal(plain[2][key], both_tails[2][key]), key
        # mode 2, the call after the next reuses a call's arrays (the library orders its K1 behind K3 of the call two back): five calls over two
        # sets of arrays, the last two equal the in-order calls
        engine.set_option("pi_defer_tail", 2)
        two = [bufs(), bufs()]
        torch.cuda.synchronize(dev)
        for k in range(5):
            engine.set_frames(batches[k % 3], None, H, W, cam, borrow=True)
            engine.processImages(N, perm, gt_jp6=gts, seed=31 + (k % 3), out=two[k & 1])
        engine.joinTail()
        engine.synchronize()
        for k in (3, 4):
            ref_call = plain[k % 3]
            for key in ("refAvgHyp", "refSteps", "out4", "inlierMaps", "avgHyp", "sfScores"):
                assert np.array_equal(ref_call[key], two[k & 1][key].cpu().numpy()), (k, key)
        # another entry point after a deferred call: ordered behind the tail without an explicit join (K7 on the refined poses of the last batch)
        engine.set_option("pi_defer_tail", 1)
        o = bufs()
        torch.cuda.synchronize(dev)
        engine.set_frames(batches[0], None, H, W, cam, borrow=True)
        engine.processImages(N, perm, gt_jp6=gts, seed=31, out=o)
        from dsac_amd.capi import lib, ptr, check
        again = np.zeros((F, 4))  # est is read on the device, the result comes back to the host (synchronous)
        check(engine._ctx, lib.dsac_loss_frames(engine._ctx, F, ptr(o["refAvgHyp"]), ptr(gts), ptr(again), None))
        assert np.array_equal(again, plain[0]["out4"])
        # a host destination: no deferral for that call, complete on return
        h = engine.processImages(N, perm, gt_jp6=gts, seed=31)
        assert np.array_equal(h["refAvgHyp"], plain[0]["refAvgHyp"]) and np.array_equal(h["out4"], plain[0]["out4"])
    finally:
        engine.set_option("pi_defer_tail", 0)


def test_deferred_tail_and_a_reused_borrowed_frame_buffer(engine, synth):
    """The deferred tail (K6 / K7) reads the frame after dsac_process_images has returned; with DSAC_FRAME_BORROW that is the caller's memory
    (include/dsac_hip.h, dsac_join_tail).  A pipeline that refills ONE borrowed coordinate buffer batch after batch must join the tail before the refill:
    then the results equal the in-order run bit for bit.  (Refilling without the join races with K6 -- the engine's stream is not ordered against
    the tail -- which is exactly why the header asks for the join.)"""
    import torch
    H, W, F, N = 120, 160, 4, 128
    P = H * W
    dev = torch.device("cuda", 0)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    srcs = []
    cam = None
    for k in range(3):
        frames = [synth.chess_like_frame(H, W, seed=400 + 10 * k + f) for f in range(F)]
        cam = frames[0]["cam"]
        srcs.append(torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev))
    gts = torch.zeros(F, 6, dtype=torch.float64, device=dev)
    from dsac_amd.capi import lib as _lib
    st = torch.cuda.ExternalStream(_lib.dsac_get_stream(engine._ctx), device=dev)

    def bufs():
        n = F * N
        return dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev),
                    ok=torch.zeros(n, dtype=torch.uint8, device=dev), scores=torch.zeros(n, dtype=torch.float64, device=dev),
                    sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(F, dtype=torch.float64, device=dev),
                    avgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev),
                    refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, dtype=torch.float64, device=dev))

    def run(defer):
        engine.set_option("pi_defer_tail", 1 if defer else 0)
        one = torch.zeros_like(srcs[0])   # the ONE borrowed buffer every batch is copied into
        outs = [bufs() for _ in range(3)]
        torch.cuda.synchronize(dev)
        for k in range(3):
            if defer:
                engine.joinTail()         # the previous batch's K6 still reads `one`: order the refill behind it
            with torch.cuda.stream(st):
                one.copy_(srcs[k])        # refill on the engine's stream
            engine.set_frames(one, None, H, W, cam, borrow=True)
            engine.processImages(N, perm, gt_jp6=gts, seed=77 + k, out=outs[k])
        engine.joinTail()
        engine.synchronize()
        return [{key: v.cpu().numpy().copy() for key, v in o.items()} for o in outs]

    try:
        plain, deferred = run(False), run(True)
        for a, b in zip(plain, deferred):
            assert (a["refSteps"] == 8).all() and a["ok"].all()
            for key in a:
                assert np.array_equal(a[key], b[key]), key
    finally:
        engine.set_option("pi_defer_tail", 0)


def test_score_tail_mode_with_small_and_large_calls_mixed(engine, synth):
    """"pi_defer_tail" = 2 puts the tail of a SMALL call (up to two full-size images' worth of pairs) on one of two alternating streams and the tail of a
    larger call on the first; a call may be given the arrays of the call two back, whichever stream that one used.  A sequence that mixes one-image and
    three-image calls at 640 x 480, arrays reused at distance two wherever the shapes agree, equals the same calls in stream order bit for bit."""
    import torch
    H, W, N = 480, 640, 256
    P = H * W
    dev = torch.device("cuda", 0)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    frames = [synth.chess_like_frame(H, W, seed=610 + f) for f in range(3)]
    cam = frames[0]["cam"]
    x3 = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
    x1 = [x3[f:f + 1].clone() for f in range(3)]
    seq = [1, 3, 1, 1, 3, 3, 1, 3, 1, 3]   # frames per call

    def bufs(F):
        n = F * N
        f64 = dict(dtype=torch.float64, device=dev)
        return dict(hyps=torch.zeros(n, 6, **f64), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev), ok=torch.zeros(n, dtype=torch.uint8, device=dev),
                    scores=torch.zeros(n, **f64), sfScores=torch.zeros(n, **f64), sfEntropy=torch.zeros(F, **f64), avgHyp=torch.zeros(F, 6, **f64),
                    refAvgHyp=torch.zeros(F, 6, **f64), refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, **f64))

    def run(mode):
        engine.set_option("pi_defer_tail", mode)
        outs = []
        pre = [bufs(F) for F in seq]
        gts = {1: torch.zeros(1, 6, dtype=torch.float64, device=dev), 3: torch.zeros(3, 6, dtype=torch.float64, device=dev)}
        torch.cuda.synchronize(dev)  # torch zero-fills on ITS stream: done before the engine's streams write the arrays; nothing below waits
        for i, F in enumerate(seq):
            # calls 8 and 9: the arrays of the call two back (same shape: allowed by the mode's contract); fresh ones otherwise
            reuse = mode == 2 and i >= 8 and seq[i - 2] == F
            o = outs[i - 2] if reuse else pre[i]
            if F == 1:
                engine.set_frames(x1[i % 3], None, H, W, cam, borrow=True)
            else:
                engine.set_frames(x3, None, H, W, cam, borrow=True)
            engine.processImages(N, perm, gt_jp6=gts[F], seed=500 + 7 * i, max_tries=1 << 16, out=o)
            outs.append(o)
        engine.joinTail()
        engine.synchronize()
        return [{k: v.cpu().numpy().copy() for k, v in o.items()} for o in outs]

    try:
        plain = run(0)
        mixed = run(2)
        # calls 8 and 9 reuse the arrays of calls 6 and 7 in mode 2: compare the calls whose arrays survive (0..5 and 8, 9)
        for i in list(range(6)) + [8, 9]:
            assert (plain[i]["refSteps"] == 8).all() and plain[i]["ok"].all()
            for key in plain[i]:
                assert np.array_equal(plain[i][key], mixed[i][key]), (i, key)
    finally:
        engine.set_option("pi_defer_tail", 0)


def test_bound_process_images_call_equals_the_generic_one(engine, synth):
    """Engine.bindProcessImages: dsac_process_images with every argument but the seed marshalled once (a loop over single images would otherwise spend as
    long in Python as on the GPU) -- the same results as the generic call, also with "device_args" set and the tails deferred."""
    import torch
    H, W, N = 120, 160, 128
    dev = torch.device("cuda", 0)
    fr = synth.chess_like_frame(H, W, seed=77)
    xyz = torch.from_numpy(fr["xyz"]).to(dev)
    perm = torch.from_numpy(synth.fast_permutations(H * W, 8)).to(dev)
    gt = torch.zeros(1, 6, dtype=torch.float64, device=dev)
    engine.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
    want = [engine.processImages(N, perm.cpu().numpy(), gt_jp6=np.zeros((1, 6)), seed=40 + i) for i in range(3)]
    f64 = dict(dtype=torch.float64, device=dev)
    outs = [dict(hyps=torch.zeros(N, 6, **f64), sampledPoints=torch.zeros(N, 4, dtype=torch.int32, device=dev), ok=torch.zeros(N, dtype=torch.uint8, device=dev),
                 scores=torch.zeros(N, **f64), sfScores=torch.zeros(N, **f64), sfEntropy=torch.zeros(1, **f64), avgHyp=torch.zeros(1, 6, **f64),
                 refAvgHyp=torch.zeros(1, 6, **f64), refSteps=torch.zeros(1, dtype=torch.int32, device=dev), out4=torch.zeros(1, 4, **f64)) for _ in range(3)]
    torch.cuda.synchronize(dev)
    calls = [engine.bindProcessImages(N, perm, o, gt_jp6=gt) for o in outs]
    try:
        engine.set_option("device_args", 1)
        engine.set_option("pi_defer_tail", 2)
        for i, c in enumerate(calls):
            c(40 + i)
        engine.joinTail()
        engine.synchronize()
    finally:
        engine.set_option("pi_defer_tail", 0)
        engine.set_option("device_args", 0)
    for w, o in zip(want, outs):
        for key in o:
            assert np.array_equal(np.asarray(w[key]).reshape(-1), o[key].cpu().numpy().reshape(-1)), key
    with pytest.raises(ValueError):
        engine.bindProcessImages(N, perm, dict(outs[0], hyps=None), gt_jp6=gt)

"""Randomised sweep over map sizes, batch sizes and thresholds: the whole test-time unit (dsac_process_images: K1, K2, K3, K6, K7) and the score
seam (dsac_process_images_begin / _finish) on shapes nobody picked by hand -- odd widths, maps that are not a multiple of any tile, one to five frames,
sampled or implicit pixel positions, float or int16-quantised coordinates.  Every case is checked three ways:
  * the batch equals the single-frame calls bit for bit (frame f = the stream of seed + f),
  * begin -> the soft-inlier sums as scores -> finish equals dsac_process_images bit for bit,
  * the oracle's chain on the same frame: identical minimal sets, error images within 1e-3 px, softmax within 1e-12 of the oracle's on the same scores,
    refinement 1e-7 / identical inlier maps, loss 1e-9 (the tolerances of SURVEY.md 8(c)).
The cases are drawn from a fixed seed: the sweep is the same on every run."""
import numpy as np
import pytest

from conftest import excl_clamp_edge, margin

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20260926)
    out = []
    for k in range(14):
        H = int(rng.integers(6, 90))
        W = int(rng.integers(6, 120))
        F = int(rng.integers(1, 6))
        N = int(rng.choice([128, 128, 256, 384]))
        out.append((k, H, W, F, N, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), float(rng.choice([5.0, 10.0, 10.0, 25.0])), int(rng.integers(1, 1 << 30))))
    return out


@pytest.mark.parametrize("k,H,W,F,N,implicit_uv,int16,thr,seed", _cases())
def test_random_shape(engine, orc, synth, k, H, W, F, N, implicit_uv, int16, thr, seed):
    P = H * W
    # implicit pixel positions are the cell indices: the camera of such a map is the full-size one scaled down, principal point at the map's centre (a
    # 640x480 camera looking at cells 0..W-1 would see a narrow off-axis window: tests/test_gpu_forward.py covers that ill-conditioned case for K1)
    cam = (525.0 * W / 640, 525.0 * W / 640, W / 2.0, H / 2.0) if implicit_uv else synth.CAM_7SCENES
    frames = [synth.chess_like_frame(H, W, seed=seed % 100000 + f, quantise_int16=int16, grid_uv=implicit_uv, cam=cam) for f in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv, cam = frames[0]["uv"], frames[0]["cam"]
    uv_arg = None if implicit_uv else uv
    steps = 8
    perm = synth.fast_permutations(P, steps)
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    max_inl, min_inl = (100, 50) if P >= 400 else (20, 8)

    def set_all():
        if F > 1:
            engine.set_frames(xyz, uv_arg, H, W, cam)
        else:
            engine.set_frame(xyz[0], uv_arg, H, W, cam)
            engine.frames = 1

    set_all()
    err = np.zeros((F * N, P), np.float32)
    b = engine.processImages(N, perm, gt_jp6=gts, seed=seed, thr=thr, err=err, want_inlier_maps=True, max_inl=max_inl, min_inl=min_inl, max_tries=4096)

    # (1) the seam with the built-in score: begin -> scores -> finish
    err2, soft = np.zeros_like(err), np.zeros(F * N)
    poses, sets, ok = engine.processImagesBegin(N, err2, seed=seed, thr=thr, soft=soft, max_tries=4096)
    r = engine.processImagesFinish(N, soft, perm, poses, gt_jp6=gts, scale=0.1, thr=thr, max_inl=max_inl, min_inl=min_inl, want_inlier_maps=True)
    assert np.array_equal(poses, b["hyps"]) and np.array_equal(sets, b["sampledPoints"]) and np.array_equal(ok, b["ok"])
    assert np.array_equal(err2, err) and np.array_equal(soft, b["scores"])
    for key in ("sfScores", "sfEntropy", "avgHyp", "refAvgHyp", "refSteps", "inlierMaps", "out4"):
        assert np.array_equal(r[key], b[key]), key

    # (2) frame by frame
    for f in range(F):
        sl = slice(f * N, (f + 1) * N)
        if F > 1:
            engine.set_frame(xyz[f], uv_arg, H, W, cam)
            engine.frames = 1
            e1 = np.zeros((N, P), np.float32)
            s = engine.processImages(N, perm, gt_jp6=gts[f:f + 1], seed=seed + f, thr=thr, err=e1, want_inlier_maps=True, max_inl=max_inl, min_inl=min_inl,
                                     max_tries=4096)
            assert np.array_equal(b["sampledPoints"][sl], s["sampledPoints"]) and np.array_equal(b["hyps"][sl], s["hyps"]) and np.array_equal(b["ok"][sl], s["ok"])
            assert np.array_equal(err[sl], e1)
            assert np.array_equal(b["scores"][sl], s["scores"]) and np.array_equal(b["sfScores"][sl], s["sfScores"]) and b["sfEntropy"][f] == s["sfEntropy"][0]
            assert np.array_equal(b["avgHyp"][f], s["avgHyp"][0]) and np.array_equal(b["refAvgHyp"][f], s["refAvgHyp"][0])
            assert np.array_equal(b["inlierMaps"][f], s["inlierMaps"][0]) and b["refSteps"][f] == s["refSteps"][0] and np.array_equal(b["out4"][f], s["out4"][0])

        # (3) the oracle on frame f
        pr, sr, okr, _ = orc.sample(N, seed + f, xyz[f], uv, H, W, cam, thr=thr, max_tries=4096)
        assert np.array_equal(okr.astype(bool), b["ok"][sl].astype(bool))
        good = b["ok"][sl].astype(bool)
        assert np.array_equal(sr[good], b["sampledPoints"][sl][good])
        hyps = b["hyps"][sl]
        ref_err = orc.get_diff_maps(hyps, xyz[f], uv, H, W, cam)
        m = excl_clamp_edge(err[sl], ref_err)
        # K2 projects in fp32: E = R X + t cancels where a scene point lies next to the camera centre (the synthetic outliers are scattered through the
        # volume, some land millimetres from it), and the residual's error grows like eps32 * f * |t| / Ez^2.  The stated 1e-3 px is for scene depth: cells
        # at least 200 mm in front of (or behind) the camera; nearer cells get the bound scaled by (200 / Ez)^2
        Ez = np.stack([(synth.rodrigues(h6[:3])[2] * xyz[f].astype(np.float64)).sum(1) + h6[5] for h6 in hyps])
        scene = np.abs(Ez) >= 200.0
        d = np.abs(err[sl] - ref_err)
        margin("a3", "random shapes: K2 residuals vs oracle on the kernel's own poses, cells at scene depth (|Ez| >= 200 mm), max px (clamp-edge cells excluded)",
               d[m & scene].max(initial=0.0), 1e-3)
        near = m & ~scene
        if near.any():
            margin("a3", "random shapes: K2 residuals vs oracle, cells within 200 mm of the camera centre: max of |d| * (Ez / 200)^2 px", (d[near] * (Ez[near] / 200.0) ** 2).max(), 1e-3)
        soft_o = orc.soft_inlier(ref_err, 10.0, 0.5)
        margin("north*", "random shapes: soft-inlier scores vs oracle, relative to the largest score", np.abs(b["scores"][sl] - soft_o).max() / max(1.0, np.abs(soft_o).max()), 1e-4)
        w_o = orc.softMax(0.1 * b["scores"][sl])
        margin("a4", "random shapes: K3 softmax on the kernel's own scores vs oracle", np.abs(w_o - b["sfScores"][sl]).max(), 1e-12)
        margin("a5", "random shapes: soft-argmax pose vs oracle on the same weights and poses", np.abs(orc.avg_pose(b["sfScores"][sl], hyps) - b["avgHyp"][f]).max(), 1e-9)
        ref_o, imap_o, sd_o = orc.refine(b["avgHyp"][f], perm, xyz[f], uv, H, W, cam, inlier_count=max_inl, min_inliers=min_inl, thr=thr, want_inlier_map=True)
        assert int(np.asarray(sd_o).reshape(-1)[0]) == int(b["refSteps"][f])
        assert np.array_equal(imap_o.reshape(-1), b["inlierMaps"][f])
        margin("a6", "random shapes: refined pose vs oracle, max-rel (inlier maps and step counts identical)",
               np.abs(np.asarray(ref_o).reshape(-1)[:6] - b["refAvgHyp"][f]).max() / max(1.0, np.abs(ref_o).max()), 1e-7)
        R1, t1 = orc.cv2our(b["refAvgHyp"][f])
        loss_o = orc.maxLoss(R1, t1, orc.rodrigues_vec2mat(gts[f][:3]), gts[f][3:])
        margin("a7", "random shapes: loss vs oracle, relative", abs(loss_o - b["out4"][f][0]) / max(1.0, loss_o), 1e-9)


def _stream_cases():
    rng = np.random.default_rng(7919)
    out = []
    for k in range(8):
        out.append((k, int(rng.integers(6, 90)), int(rng.integers(6, 140)), int(rng.integers(1, 7)), int(rng.choice([128, 256])), int(rng.choice([1, 3, 8])),
                    int(rng.choice([1, 2])), bool(rng.integers(0, 2)), int(rng.integers(1, 1 << 30))))
    return out


@pytest.mark.parametrize("k,H,W,F,N,steps,mode,copy_frames,seed", _stream_cases())
def test_random_shape_as_a_stream_of_batches(synth, k, H, W, F, N, steps, mode, copy_frames, seed):
    """Four batches of different frames in a row with the tails deferred ("pi_defer_tail" 1 / 2), every buffer on the device, the frames either borrowed
    or COPIED by dsac_set_frames right behind a call whose tail still reads the previous ones (the copy must wait for it): equal to the same four calls
    on host arrays in stream order, bit for bit -- on random map sizes, batch sizes and refinement step counts (1, 3, 8)."""
    import torch
    import dsac_amd
    P = H * W
    dev = torch.device("cuda", 0)
    cam = synth.CAM_7SCENES
    perm = synth.fast_permutations(P, steps)
    max_inl, min_inl = (100, 50) if P >= 400 else (20, 8)
    batches = []
    for b in range(4):
        frames = [synth.chess_like_frame(H, W, seed=seed % 100000 + 10 * b + f) for f in range(F)]
        batches.append(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames])))
    uv = synth.chess_like_frame(H, W, seed=1)["uv"]
    gts = np.zeros((F, 6))
    gts[:, 5] = 2500.0
    keys = ("hyps", "sampledPoints", "ok", "scores", "sfScores", "sfEntropy", "avgHyp", "refAvgHyp", "refSteps", "out4", "inlierMaps")

    with dsac_amd.Engine(0) as e:
        plain = []
        for b in range(4):
            e.set_frames(batches[b], uv, H, W, cam)
            plain.append(e.processImages(N, perm, gt_jp6=gts, seed=seed + b, want_inlier_maps=True, max_inl=max_inl, min_inl=min_inl, max_tries=4096))

    def bufs():
        n = F * N
        return dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev),
                    ok=torch.zeros(n, dtype=torch.uint8, device=dev), scores=torch.zeros(n, dtype=torch.float64, device=dev),
                    sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(F, dtype=torch.float64, device=dev),
                    avgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev),
                    refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, dtype=torch.float64, device=dev),
                    inlierMaps=torch.zeros(F, P, dtype=torch.int32, device=dev))

    with dsac_amd.Engine(0) as e:
        e.set_option("pi_defer_tail", mode)
        perm_d, gts_d, uv_d = torch.from_numpy(perm).to(dev), torch.from_numpy(gts).to(dev), torch.from_numpy(uv).to(dev)
        xyz_d = [torch.from_numpy(x).to(dev) for x in batches]
        outs = [bufs() for _ in range(4)]
        torch.cuda.synchronize(dev)
        for b in range(4):
            # copy_frames: the context copies the coordinates into its own buffer -- behind the tail of the previous call, which reads that buffer
            e.set_frames(xyz_d[b], uv_d, H, W, cam, borrow=not copy_frames)
            e.processImages(N, perm_d, gt_jp6=gts_d, seed=seed + b, out=outs[b], max_inl=max_inl, min_inl=min_inl, max_tries=4096)
        e.joinTail()
        e.synchronize()
        for b in range(4):
            for key in keys:
                assert np.array_equal(plain[b][key], outs[b][key].cpu().numpy()), (b, key)

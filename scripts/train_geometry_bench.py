"""The training path's GEOMETRY as the device sees it: every argument and result lives in HBM (torch tensors through the C ABI), nothing
is copied, the host only enqueues -- the chain of one step is

  dsac_process_images (K1 + K2 + K3 -> K6 refine + inlier map -> K7 loss)  ->  dsac_backward_path1 (K7 backward, the 12 + 6n finite-difference
  replicas of K6, gradient assembly, K5 dPNP, softmax backward)  ->  K4 fused soft-inlier backward (dsac_soft_score_backward)

on reference-sized 40x40 int16 frames and on 640x480 frames, 256 hypotheses each, for F = 1, 8 and 16 frames per step (round 4: every backward
stage takes a frame batch -- one launch per stage for all frames, one gradient per frame).  core/train_ransac_softam.cpp:288-394 trains on one
image per round; F frames per step is what a data-parallel step puts on one GPU.
scripts/train_path_profile.py runs the same functions with numpy arguments (host copies and a synchronisation per call); this is what the end-to-end
trainer (dsac_amd/e2e.py) pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dsac_amd
from dsac_amd import synth
from dsac_amd.capi import lib, ptr, check

dev = torch.device("cuda:0")
eng = dsac_amd.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
ctx = eng._ctx
N = 256
ONLY = os.environ.get("DSAC_TGB_ONLY")  # e.g. "480x640x16": one configuration (for rocprofv3 --stats of exactly that chain)
for (H, W, reps) in ((40, 40, 200), (480, 640, 100)):
    P = H * W
    for F in (1, 8, 16):
        if ONLY and ONLY != "%dx%dx%d" % (H, W, F):
            continue
        frs = [synth.chess_like_frame(H, W, seed=1305 + f, quantise_int16=(H == 40)) for f in range(F)]
        xyz = torch.as_tensor(np.ascontiguousarray(np.stack([f_["xyz"] for f_ in frs])), device=dev)
        uv = torch.as_tensor(frs[0]["uv"], device=dev) if H == 40 else None
        eng.set_frames(xyz, uv, H, W, frs[0]["cam"], borrow=True) if F > 1 else eng.set_frame(xyz[0], uv, H, W, frs[0]["cam"], borrow=True)
        perm = torch.as_tensor(synth.fast_permutations(P, 8), device=dev)
        gt = torch.as_tensor(np.stack([synth.cv_to_jp6(f_["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for f_ in frs]), device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        NF = N * F
        o = dict(hyps=torch.zeros(NF, 6, **f64), sampledPoints=torch.zeros(NF, 4, dtype=torch.int32, device=dev), ok=torch.zeros(NF, dtype=torch.uint8, device=dev),
                 scores=torch.zeros(NF, **f64), sfScores=torch.zeros(NF, **f64), sfEntropy=torch.zeros(F, **f64), avgHyp=torch.zeros(F, 6, **f64),
                 refAvgHyp=torch.zeros(F, 6, **f64), refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, **f64),
                 inlierMaps=torch.zeros(F, P, dtype=torch.int32, device=dev))
        g, dpnp, grad = torch.zeros(NF, **f64), torch.zeros(NF, 6, 12, **f64), torch.zeros(F * P, 3, **f64)
        torch.cuda.synchronize()

        def step(seed):
            eng.processImages(N, perm, gt_jp6=gt, seed=seed, scale=0.1, out=o)
            grad.zero_()  # torch's current stream IS the engine's stream (set_stream above): ordered, no synchronisation
            check(ctx, lib.dsac_backward_path1(ctx, NF, ptr(o["hyps"]), ptr(o["sampledPoints"]), ptr(o["sfScores"]), ptr(o["avgHyp"]), ptr(o["refAvgHyp"]), ptr(gt),
                                               ptr(perm), 8, 100, 50, 10.0, ptr(o["inlierMaps"]), 0.01, 0.001, 2.0, 0.1, ptr(dpnp), ptr(grad), ptr(g), None, None))
            eng.dSoftScore(o["hyps"], o["sampledPoints"], g, dpnp=dpnp, grad=grad)

        for i in range(3):
            step(7 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            step(100 + 16 * i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("device-resident training geometry %dx%d, N=%d, %2d frame(s) per step: %7.1f us per step = %6.1f us per frame (enqueue-only host, one synchronisation "
              "per %d steps), mean loss %.3f, accepted %d of %d, refinement steps >= %d" %
              (W, H, N, F, dt * 1e6, dt * 1e6 / F, reps, float(o["out4"][:, 0].mean().item()), int(o["ok"].sum().item()), NF, int(o["refSteps"].min().item())))

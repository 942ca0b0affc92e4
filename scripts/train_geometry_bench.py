"""The training path's GEOMETRY as the device sees it: every argument and result lives in HBM (torch tensors through the C ABI), nothing
is copied, the host only enqueues -- one frame's chain is

  K1 + K2 + K3 (dsac_score_hypotheses, soft-inlier score) -> K6 refine + inlier map -> K7 loss -> dsac_backward_path1 (K7 backward, the
  12 + 6n finite-difference replicas of K6, gradient assembly, K5 dPNP, softmax backward) -> K4 fused soft-inlier backward

on a reference-sized 40x40 int16 frame and on a 640x480 frame, 256 hypotheses.  scripts/train_path_profile.py runs the same functions with
numpy arguments (host copies and a synchronisation per call); this is what the end-to-end trainer (dsac_amd/e2e.py) pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dsac_amd
from dsac_amd import synth
from dsac_amd.capi import lib, ptr, check
from oracle import oracle as orc  # cv_to_jp6 only (pose convention of the ground truth)

dev = torch.device("cuda:0")
eng = dsac_amd.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
ctx = eng._ctx
N = 256
for (H, W, reps) in ((40, 40, 200), (480, 640, 100)):
    P = H * W
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40))
    xyz = torch.as_tensor(fr["xyz"], device=dev)
    uv = torch.as_tensor(fr["uv"], device=dev) if H == 40 else None
    eng.set_frame(xyz, uv, H, W, fr["cam"], borrow=True)
    perm = torch.as_tensor(synth.fast_permutations(P, 8), device=dev)
    gt = torch.as_tensor(orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])), device=dev)
    f64 = dict(dtype=torch.float64, device=dev)
    poses, sets, ok = torch.zeros(N, 6, **f64), torch.zeros(N, 4, dtype=torch.int32, device=dev), torch.zeros(N, dtype=torch.uint8, device=dev)
    scores, w, ent, avg = torch.zeros(N, **f64), torch.zeros(N, **f64), torch.zeros(1, **f64), torch.zeros(6, **f64)
    ref, out4, g, dpnp, grad = torch.zeros(6, **f64), torch.zeros(4, **f64), torch.zeros(N, **f64), torch.zeros(N, 6, 12, **f64), torch.zeros(P, 3, **f64)
    imap, sd = torch.zeros(P, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def frame(seed):
        eng.scoreHypotheses(N, seed=seed, scale=0.1, out=(poses, sets, ok, scores, w, ent, avg))
        imap.zero_()  # torch's current stream IS the engine's stream (set_stream below): ordered, no synchronisation
        grad.zero_()
        check(ctx, lib.dsac_refine(ctx, 1, ptr(avg), ptr(perm), 8, 100, 50, 10.0, None, None, ptr(ref), ptr(imap), ptr(sd)))
        check(ctx, lib.dsac_loss(ctx, ptr(ref), ptr(gt), ptr(out4), None))
        check(ctx, lib.dsac_backward_path1(ctx, N, ptr(poses), ptr(sets), ptr(w), ptr(avg), ptr(ref), ptr(gt), ptr(perm), 8, 100, 50, 10.0, ptr(imap),
                                           0.01, 0.001, 2.0, 0.1, ptr(dpnp), ptr(grad), ptr(g), None, None))
        eng.dSoftScore(poses, sets, g, dpnp=dpnp, grad=grad)

    for i in range(3):
        frame(7 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        frame(100 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("device-resident training geometry %dx%d, N=%d: %.0f us per frame (enqueue-only host, one synchronisation per %d frames), loss %.3f, accepted %d" %
          (W, H, N, dt * 1e6, reps, float(out4[0].item()), int(ok.sum().item())))

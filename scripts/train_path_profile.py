"""The training path of one frame on the engine alone (no CNNs): processImage forward + the trainer's backward section
(train_ransac_softam.cpp:288-394) on a reference-sized 40x40 frame and on a 640x480 frame, then the DSAC variant (all hypotheses refined,
batched dRefine).  Meant to run under `rocprofv3 --kernel-trace --stats`: the per-kernel table is the evidence for K1, K3, K5, K6
(single problem and the 12 + 6n finite-difference batch), K7 and the DSAC-variant launches (profiles/r02_train_path_kernel_stats.csv)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth

eng = dsac_amd.Engine(0)
for (H, W, N, reps) in ((40, 40, 256, 20), (480, 640, 256, 5)):
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40))
    eng.set_frame(fr["xyz"], fr["uv"] if H == 40 else None, H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8)
    gt = synth.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
    t0 = time.perf_counter()
    for i in range(reps):
        fwd = eng.processImage(N=N, seed=1305 + i, perm=perm, gt_jp6=gt)
        bwd = eng.backward(fwd, gt)
    dt = (time.perf_counter() - t0) / reps
    print("soft-argmax training path %dx%d, N=%d: %.2f ms per frame (host-synchronous calls), loss %.3f, refine steps %d, |grad| %.3e" %
          (W, H, N, dt * 1e3, fwd["loss"], fwd["refSteps"], np.linalg.norm(bwd["grad"])))
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
eng.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
perm = synth.fast_permutations(1600, 8)
gt = synth.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
t0 = time.perf_counter()
for i in range(10):
    f = eng.processImageDSAC(N=256, seed=7 + i, perm=perm, gt_jp6=gt, draw_u=0.5)
    b = eng.backwardDSAC(f, gt)
print("DSAC variant 40x40, N=256: %.2f ms per frame, expected loss %.3f" % ((time.perf_counter() - t0) / 10 * 1e3, f["expectedLoss"]))

"""K6 (k_refine) alone: one refinement problem (the forward of processImage, cnn_softam.h:1099-1154) and a batch of 256 problems (the DSAC
variant's all-hypotheses refinement, cnn.h:1155-1215) on a 40x40 and a 640x480 frame.  Run under `rocprofv3 --kernel-trace --stats`;
the host clock printed here includes the numpy argument copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = dsac_amd.Engine(0)
rng = np.random.default_rng(5)
for (H, W) in ((40, 40), (480, 640)):
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40))
    eng.set_frame(fr["xyz"], fr["uv"] if H == 40 else None, H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8)
    init = fr["gt_pose"][None, :] + rng.normal(size=(B, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0])
    eng.refine(init, perm)
    t0 = time.perf_counter()
    for i in range(reps):
        poses, sd = eng.refine(init, perm)
    dt = (time.perf_counter() - t0) / reps
    d = poses - fr["gt_pose"][None, :]
    print("k_refine %dx%d  B=%d: %.1f us per call (host clock), steps done %.2f avg, |rvec - gt| %.2e  |t - gt| %.3f mm  checksum %.12e" %
          (W, H, B, dt * 1e6, sd.mean(), np.abs(d[:, :3]).max(), np.abs(d[:, 3:]).max(), float(poses.sum())))

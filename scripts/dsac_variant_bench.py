"""Wall time of the DSAC-variant forward (all N hypotheses refined) and of its training backward on the reference-sized 40 x 40 map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
eng = dsac_amd.Engine(0)
eng.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
perm = synth.fast_permutations(1600, 8)
R = synth.rodrigues(fr["gt_pose"][:3]); F = np.diag([1.0, -1.0, -1.0]); Rj = F @ R; tj = F @ fr["gt_pose"][3:]
th = np.arccos(np.clip((np.trace(Rj) - 1) / 2, -1, 1)); ax = np.array([Rj[2, 1] - Rj[1, 2], Rj[0, 2] - Rj[2, 0], Rj[1, 0] - Rj[0, 1]])
gt = np.concatenate([ax / (2 * np.sin(th)) * th, tj + np.array([5.0, -8.0, 12.0])])
for N in (64, 256):
    fwd = eng.processImageDSAC(N=N, seed=1, perm=perm, gt_jp6=gt)
    bwd = eng.backwardDSAC(fwd, gt)
    t0 = time.perf_counter()
    for i in range(5): fwd = eng.processImageDSAC(N=N, seed=2 + i, perm=perm, gt_jp6=gt)
    t1 = time.perf_counter()
    for i in range(5): bwd = eng.backwardDSAC(fwd, gt)
    t2 = time.perf_counter()
    nsel = int((fwd["sfScores"] > 1e-4).sum())
    print("DSAC variant N=%3d: forward (sample, score, softmax, refine all %d, losses) %.2f ms; backward (%d weighted hypotheses, dRefine batch + dSMScore) %.2f ms; expected loss %.3f"
          % (N, N, (t1 - t0) / 5 * 1e3, nsel, (t2 - t1) / 5 * 1e3, fwd["expectedLoss"]))

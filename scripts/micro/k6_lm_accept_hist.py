"""how often does CvLevMarq (K6's LM chain, core/cnn_softam.h:1099-1154 -> solvePnP ITERATIVE) REJECT a trial step?  A rejected step costs one more
dependent solve + residual pass on the one wave that is the latency of an image.  Needs the instrumented build:  make -C dsac_amd/csrc lmstats ;
DSAC_HIP_LIB=dsac_amd/csrc/build/ab/libdsac_hip_lmstats.so python scripts/micro/k6_lm_accept_hist.py
(steps_done then carries accepted << 8 | rejected << 20 for the whole refinement of 8 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dsac_amd
from dsac_amd import synth

eng = dsac_amd.Engine(0)
rng = np.random.default_rng(5)
rows = []
cases = [(480, 640, 0.3, 64, "benchmark frame (70 % inliers), refinements from the soft-argmax pose +- noise"),
         (480, 640, 0.9, 64, "hard frame (10 % inliers)"), (40, 40, 0.3, 64, "40 x 40 sub-sampled, int16 (the reference's size)"), (40, 40, 0.6, 64, "40 x 40, 40 % inliers")]
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
print("# accepted / rejected Levenberg-Marquardt trial steps per refinement (8 refinement steps, up to 20 LM iterations each), instrumented K6")
for (H, W, outl, B, what) in cases:
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40), outlier_frac=outl)
    eng.set_frame(fr["xyz"], fr["uv"] if H == 40 else None, H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8)
    init = fr["gt_pose"][None, :] + rng.normal(size=(B, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0])
    out, sd = eng.refine(init, perm, max_inl=100, min_inl=50, thr=10.0)
    sd = np.asarray(sd).astype(np.int64)
    steps, acc, rej = sd & 0xff, (sd >> 8) & 0xfff, (sd >> 20) & 0x7ff
    tot = acc + rej
    print("%-82s refinement steps %.2f | accepted %.1f, rejected %.2f per refinement = %.1f %% of the trial steps rejected (max %d of %d in one refinement)" %
          (what, steps.mean(), acc.mean(), rej.mean(), 100.0 * rej.sum() / max(1, tot.sum()), rej.max(), tot[np.argmax(rej)]))
    print("    histogram of rejected steps per refinement: %s" % dict(zip(*[a.tolist() for a in np.unique(rej, return_counts=True)])))
for v in (1, 2):
    g = np.load(os.path.join(gold, "ref_frame_v%d.npz" % v))
    eng.set_frame(g["estObj"].astype(np.float32), g["sampling"].astype(np.float32), 40, 40, g["cam"])
    out, sd = eng.refine(g["avgHyp"][None, :], g["pixelIdxs"], max_inl=100, min_inl=50, thr=10.0)
    sd = int(np.asarray(sd).reshape(-1)[0])
    print("golden frame v%d (the real reference's soft-argmax pose and shuffles): steps %d, accepted %d, rejected %d" % (v, sd & 0xff, (sd >> 8) & 0xfff, (sd >> 20) & 0x7ff))
eng.close()

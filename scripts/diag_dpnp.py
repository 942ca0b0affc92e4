import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
eng = dsac_amd.Engine(0)
eng.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
for seed in (21, 22, 23):
    poses, sets, ok, _ = orc.sample(64, seed, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    J = eng.dPNP(sets, eps=0.1)
    rel = np.zeros(64); noise = np.zeros(64)
    for h in range(64):
        Jr = orc.dPNP(fr["uv"][sets[h]], fr["xyz"][sets[h]], fr["cam"], eps=0.1)
        sc = max(1.0, np.abs(Jr).max())
        rel[h] = np.abs(J[h] - Jr).max() / sc
        X0 = fr["xyz"][sets[h]].copy()
        for c in range(12):
            X1 = X0.copy().reshape(-1)
            X1[c] = np.nextafter(X1[c], np.float32(1e9))
            J2 = orc.dPNP(fr["uv"][sets[h]], X1.reshape(4, 3), fr["cam"], eps=0.1)
            noise[h] = max(noise[h], np.abs(J2 - Jr).max() / sc)
    o = np.argsort(-rel)[:6]
    print("seed", seed, "median %.2e p90 %.2e max %.2e" % (np.median(rel), np.quantile(rel, .9), rel.max()))
    for h in o: print("   h %2d rel %.2e  one-ulp sensitivity %.2e  |J|max %.2e" % (h, rel[h], noise[h], np.abs(J[h]).max()))

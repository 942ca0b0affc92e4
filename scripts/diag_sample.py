import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
for (H, W, seed, q) in ((40, 40, 1305, True), (480, 640, 99, False)):
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=q)
    eng = dsac_amd.Engine(0)
    eng.set_frame(fr["xyz"], fr["uv"], H, W, fr["cam"])
    pr, sr, okr, tries = orc.sample(256, seed, fr["xyz"], fr["uv"], H, W, fr["cam"], max_tries=4096)
    pg, sg, okg = eng.sample(256, seed=seed, max_tries=4096)
    d = np.abs(pg - pr) / (np.abs(pr) + 1e-3)
    worst = np.argsort(-d.max(1))[:5]
    print(H, W, "sets equal", np.array_equal(sg, sr), "max rel", d.max(), "median rel", np.median(d.max(1)))
    for h in worst:
        print("  h", h, "rel", d[h].max(), "gpu", pg[h], "cpu", pr[h])
        e = orc.get_diff_maps(np.stack([pg[h], pr[h]]), fr["xyz"][sr[h]], fr["uv"][sr[h]], 1, 4, fr["cam"])
        print("     4pt residuals gpu", e[0], "cpu", e[1])
    eng.close()

import numpy as np, sys
sys.path.insert(0, '.')
import dsac_amd
from dsac_amd import synth
from dsac_amd.synth import rodrigues
from oracle import oracle as orc
def dev(pg, pr):
    ang = np.array([np.degrees(np.arccos(np.clip((np.trace(rodrigues(a[:3]) @ rodrigues(b[:3]).T) - 1) / 2, -1, 1))) for a, b in zip(pg, pr)])
    trel = np.linalg.norm(pg[:, 3:] - pr[:, 3:], axis=1) / np.maximum(np.linalg.norm(pr[:, 3:], axis=1), 1e-9)
    return ang, trel
e = dsac_amd.Engine(0)
for name, fr, seed in (("40x40", synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True), 1305), ("640x480", synth.chess_like_frame(480, 640, seed=1305), 99)):
    H, W = fr["H"], fr["W"]
    e.set_frame(fr["xyz"], fr["uv"], H, W, fr["cam"])
    pr, sr, okr, _ = orc.sample(256, seed, fr["xyz"], fr["uv"], H, W, fr["cam"], thr=10.0, max_tries=4096)
    for horn in (0, 1):
        e.set_option("k1_horn", horn)
        pg, sg, okg = e.sample(256, seed=seed, thr=10.0, max_tries=4096)
        ang, trel = dev(pg, pr)
        tight = (ang <= 1e-5) & (trel <= 1e-6)
        print(name, "horn", horn, "sets equal", np.array_equal(sg, sr), "tight %.3f" % tight.mean(), "max ang %.3g deg" % ang.max(), "n>1e-3deg", int((ang > 1e-3).sum()), "worst", np.argsort(-ang)[:4].tolist())
    # conditioning of the worst ones: the ORACLE against itself on inputs moved by one float ulp
    worst = np.argsort(-ang)[:4]
    for h in worst:
        X = fr["xyz"][sr[h]].copy(); uv = fr["uv"][sr[h]]
        ok0, p0 = orc.solve_p3p(X, uv, fr["cam"])
        X2 = X.copy(); X2[0, 0] = np.nextafter(X2[0, 0], np.float32(1e9))
        ok1, p1 = orc.solve_p3p(X2, uv, fr["cam"])
        a, t = dev(p0[None], p1[None])
        print("   hyp %d: engine-vs-oracle %.3g deg; oracle vs oracle with X[0,0] moved by one ulp: %.3g deg (%s %s)" % (h, ang[h], a[0], ok0, ok1))

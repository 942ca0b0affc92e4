import numpy as np, sys
sys.path.insert(0, '.')
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
N = 300
e = dsac_amd.Engine(0)
e.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
poses, sets, ok, _ = orc.sample(N, 6, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
rng = np.random.default_rng(11)
d_err = rng.normal(size=(N, 1600)).astype(np.float32)
d_err[np.arange(N)[:, None], sets] = 0
ref, G6, S = orc.dScore(sets, d_err.astype(np.float64), fr["xyz"], fr["uv"], 40, 40, fr["cam"])
J = np.stack([orc.dPNP(fr["uv"][s], fr["xyz"][s], fr["cam"]) for s in sets])
sup = np.zeros(1600, bool); sup[sets.ravel()] = True
res = {}
for v in (0, 1, 2):
    e.set_option("k4_variant", v)
    got = e.dScore(poses, sets, d_err, dpnp=J.reshape(N, 72))
    pg = e.lastPoseGradients(N)
    d = np.abs(got - ref)
    p = np.unravel_index(d.argmax(), d.shape)
    relg = np.abs(pg - G6).max(1) / np.abs(G6).max(1)
    print("variant", v, "max err %.3e at pixel %d (support: %s) ref max %.3e | non-support max err %.3e | G6 rel: median %.2e max %.2e (hyp %d) | max|J| of worst hyp %.2e" %
          (d.max(), p[0], sup[p[0]], np.abs(ref).max(), d[~sup].max(), np.median(relg), relg.max(), relg.argmax(), np.abs(J[relg.argmax()]).max()))
    # which hypotheses own the worst pixel
    owners = np.flatnonzero((sets == p[0]).any(1))
    print("   owners of worst pixel:", owners[:6], "their G6 rel", relg[owners][:6], "max|J|", [float(np.abs(J[o]).max()) for o in owners[:6]])
    res[v] = got
print("v1 vs v0 max diff %.3e, v2 vs v0 %.3e, v2 vs v1 %.3e" % (np.abs(res[1] - res[0]).max(), np.abs(res[2] - res[0]).max(), np.abs(res[2] - res[1]).max()))

#!/usr/bin/env python3
"""Round 4 diagnosis: the fused soft-inlier form of K4 at N = 256 x 640x480 -- which hypotheses' pose sums differ from the oracle's, and is it the
in-kernel weight (sigmoid' of the fp32 residual) or the sums?  The same hypotheses go through (a) dsac_soft_score_backward and (b)
dsac_score_backward fed with the ORACLE's soft gradient images rounded to float32."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
orc.set_num_threads(orc.effective_cpus()[0])
H, W, N, TAU, BETA = 480, 640, 256, 10.0, 0.5
P = H * W
fr = synth.chess_like_frame(H, W, seed=1305)
uv = synth.pixel_grid(H, W)
poses, sets, ok, _ = orc.sample(N, 7, fr["xyz"], uv, H, W, fr["cam"])
g = np.random.default_rng(3).normal(size=N)
err = orc.get_diff_maps(poses, fr["xyz"], uv, H, W, fr["cam"]).astype(np.float64)
s = 1.0 / (1.0 + np.exp(-BETA * (TAU - err)))
dd = g[:, None] * (-BETA) * s * (1 - s)
rows_ = np.arange(N)[:, None]
dd[rows_, sets] = np.where(err[rows_, sets] < 1e-3, 0.0, dd[rows_, sets])  # the three P3P points only: the fourth point of a set is an ordinary cell
soft_scores = s.sum(1)
G6o = np.zeros((N, 6))
grad_o = np.zeros((P, 3))
for a in range(0, N, 64):
    grad_o, g6, _ = orc.dScore(sets[a:a + 64], dd[a:a + 64], fr["xyz"], uv, H, W, fr["cam"], grad=grad_o)
    G6o[a:a + 64] = g6
dpnp = np.stack([orc.dPNP(uv[s_], fr["xyz"][s_], fr["cam"]) for s_ in sets])
with dsac_amd.Engine(0) as eng:
    eng.set_frame(fr["xyz"], None, H, W, fr["cam"])
    eng.dSoftScore(poses, sets, g, tau=TAU, beta=BETA, dpnp=dpnp)
    Ga = eng.lastPoseGradients(N)
    eng.dScore(poses, sets, dd.astype(np.float32), dpnp=dpnp)
    Gb = eng.lastPoseGradients(N)
    # (c) the d_err form fed with weights formed in float32 from the ENGINE's own fp32 error images (K2): is the fused form's deviation the fp32 residual?
    e32 = eng.getDiffMap(poses).reshape(N, P)
    s32 = (np.float32(1) / (np.float32(1) + np.exp(np.float32(-BETA) * (np.float32(TAU) - e32)))).astype(np.float32)
    d32 = (g[:, None].astype(np.float32) * np.float32(-BETA) * s32 * (np.float32(1) - s32)).astype(np.float32)
    d32[rows_, sets] = np.where(err[rows_, sets] < 1e-3, np.float32(0), d32[rows_, sets])
    eng.dScore(poses, sets, d32, dpnp=dpnp)
    Gc = eng.lastPoseGradients(N)
    Gv = {}
    for v in (0, 1, 2, 3):
        eng.set_option("k4_variant", v)
        eng.dSoftScore(poses, sets, g, tau=TAU, beta=BETA, dpnp=dpnp)
        Gv[v] = eng.lastPoseGradients(N)
    eng.set_option("k4_variant", -1)
    print("max |err_K2 - err_oracle| over all cells (clamp edge included): %.3e" % np.abs(e32 - err).max())
sc = np.abs(G6o).max(1)
ra, rb = np.abs(Ga - G6o).max(1) / sc, np.abs(Gb - G6o).max(1) / sc
print("fused soft : pose sums max-rel: median %.2e p95 %.2e max %.2e" % (np.median(ra), np.quantile(ra, 0.95), ra.max()))
print("d_err form fed with the oracle's soft gradient images: median %.2e p95 %.2e max %.2e" % (np.median(rb), np.quantile(rb, 0.95), rb.max()))
rc = np.abs(Gc - G6o).max(1) / sc
print("d_err form fed with fp32 weights from K2's own error images: median %.2e p95 %.2e max %.2e" % (np.median(rc), np.quantile(rc, 0.95), rc.max()))
for v, G in Gv.items():
    r = np.abs(G - G6o).max(1) / sc
    print("fused soft, k4_variant %d: median %.2e p95 %.2e max %.2e (vs variant -1: max |d| / scale %.2e)" % (v, np.median(r), np.quantile(r, 0.95), r.max(), (np.abs(G - Ga).max(1) / sc).max()))
for h in np.argsort(-ra)[:8]:
    print("hyp %3d rel %.2e (d_err form %.2e)  soft score %.1f  g %.3f  scale %.3e  inliers<tau %d\n   oracle %s\n   fused  %s\n   (c)    %s" %
          (h, ra[h], rb[h], soft_scores[h], g[h], sc[h], (err[h] < TAU).sum(), np.array2string(G6o[h], precision=4), np.array2string(Ga[h], precision=4),
           np.array2string(Gc[h], precision=4)))
print("correlation of the error with the soft score: rank corr %.2f" % np.corrcoef(np.argsort(np.argsort(ra)), np.argsort(np.argsort(soft_scores)))[0, 1])

#!/usr/bin/env python3
"""Round 4, VERDICT item 4: error-images-only K2 at configs[2] (N = 4096 random poses, 640x480) -- which form, and does pacing the stores WITHOUT the
sigmoid arithmetic (k2_flags bits 16-23: the wave idles after a chunk's four stores) reach the rate of the err + soft launch?  Every candidate is
timed with the dispatch-attached events of dsac_profile_enable, alternating with the err + soft reference on the same box.
usage: r04_k2_err_ab.py [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402

N, H, W = 4096, 480, 640
P = H * W
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
poses = torch.from_numpy(synth.random_poses(N, seed=7) + np.array([0, 0, 0, 0, 0, 2500.0])).to(dev)
err = torch.zeros(N, P, dtype=torch.float32, device=dev)
soft = torch.zeros(N, dtype=torch.float64, device=dev)
eng.profile_enable(True, stride=1)
ABYTES = 12 * P + 48 * N + 4 * N * P + 4 * N


def run(variant, flags, order, with_soft, reps=8):
    eng.set_option("k2_variant", variant)
    eng.set_option("k2_flags", flags)
    eng.set_option("k2_order", order)
    for _ in range(2):
        eng.reproject(poses, N=N, clamp=100.0, err=err, soft=soft if with_soft else None, tau=10.0, beta=0.5)
    eng.synchronize()
    eng.profile_read(0, reset=True)
    for _ in range(reps):
        eng.reproject(poses, N=N, clamp=100.0, err=err, soft=soft if with_soft else None, tau=10.0, beta=0.5)
    eng.synchronize()
    ms, n = eng.profile_read(0, reset=True)
    return ms / n * 1e3


cands = [("both (err + soft), auto = form 58 plain order", -1, 0, 1, True),
         ("err only, auto (round 4: the fused kernel, sums dropped)", -1, 0, 1, False),
         ("err only, VALU form HT 32 (round-3 policy)", 0, 1 << 24, 1, False),
         ("err only, streaming form 58 without the sigmoid", 58, 32, 1, False),
         ("err only, streaming form 45 without the sigmoid", 45, 32, 1, False)]
for v in (58, 45):
    for ps, pn in ((0, 2), (1, 0), (2, 0), (3, 0), (4, 0), (6, 0)):
        cands.append(("err only, form %d idling %d x 64 + %d x 16 clk per chunk" % (v, ps, pn), v, 32 | (ps << 16) | (pn << 20), 1, False))
res = {c[0]: [] for c in cands}
for _ in range(40):  # settle clocks
    eng.reproject(poses, N=N, clamp=100.0, err=err, soft=soft, tau=10.0, beta=0.5)
eng.synchronize()
for r in range(rounds):
    order_ = cands if r % 2 == 0 else cands[::-1]  # alternate the order: drift of the box's clocks must not favour a candidate
    for name, v, fl, order, ws in order_:
        res[name].append(run(v, fl, order, ws))
print("K2 at N = %d x %dx%d (BASELINE configs[2]), %d rounds x 8 launches per candidate (order alternating), us per launch: min / median / max, fraction of 8 TB/s at the median" % (N, W, H, rounds))
for name, *_ in cands:
    a = np.array(res[name])
    print("%-64s %7.1f %7.1f %7.1f   %.3f" % (name, a.min(), np.median(a), a.max(), ABYTES / (np.median(a) * 1e-6) / 8e12))
eng.close()

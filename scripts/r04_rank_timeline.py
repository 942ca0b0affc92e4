"""Kernel timeline of ONE emulated rank of 8 (dsac_amd.shard.ShardRunner: 8 images x 256 hypotheses x 640x480 per step, refinement tail deferred across steps,
one-step-late exchange on a side stream).  Run under `rocprofv3 --kernel-trace`, then `python scripts/r04_rank_timeline.py <kernel_trace.csv>` prints, over
the last 40 steps: the period of the K2 launches (= the step), K2's own duration, the gap between the end of a K2 and the start of the next K1, how much of
every K6 / K7 launch ran under a K1 / K2 of the NEXT step, and what else sat on the engine's stream between two K2 launches."""
import csv
import sys

if len(sys.argv) > 1:
    rows = sorted(((r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[1])
    k2 = [(s, e) for n, s, e in rows if "k_reproject" in n][-41:]
    k1 = [(s, e) for n, s, e in rows if "k_sample" in n][-41:]
    k6 = [(s, e) for n, s, e in rows if "k_refine" in n][-40:]
    k7 = [(s, e) for n, s, e in rows if "k_pose_loss" in n][-40:]
    per = [(b[0] - a[0]) / 1e3 for a, b in zip(k2[:-1], k2[1:])]
    dur = [(e - s) / 1e3 for s, e in k2[1:]]
    k1d = [(e - s) / 1e3 for s, e in k1[1:]]
    gap = []
    for (s2, e2) in k2[:-1]:
        nxt = [s for s, e in k1 if s >= e2]
        if nxt:
            gap.append((nxt[0] - e2) / 1e3)
    big = k1 + k2

    def under(ks):
        tot = ov = 0
        for s, e in ks:
            tot += e - s
            ov += sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in big)
        return tot / max(1, len(ks)) / 1e3, 100.0 * ov / max(1, tot)
    m6, o6 = under(k6)
    m7, o7 = under(k7)
    import statistics as st
    print("last %d steps of the emulated rank: step period (K2 start to K2 start) median %.1f us (min %.1f, max %.1f)" % (len(per), st.median(per), min(per), max(per)))
    print("  K2 (8 frames) median %.1f us; K1 median %.1f us; end of K2 -> start of the next step's K1: median %.1f us" % (st.median(dur), st.median(k1d), st.median(gap)))
    print("  K6: mean %.1f us per launch, %.0f %% of it under a K1 / K2 launch of the next step; K7: mean %.1f us, %.0f %% under K1 / K2" % (m6, o6, m7, o7))
    names = {}
    lo, hi = k2[-11][1], k2[-1][0]
    for n, s, e in rows:
        if lo <= s <= hi and "k_reproject" not in n:
            d = names.setdefault(n[:70], [0, 0.0])
            d[0] += 1
            d[1] += (e - s) / 1e3
    print("  kernels between the last 10 K2 launches (count per 10 steps, us per step):")
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print("    %-70s %4d  %7.1f" % (n, c, t / 10))
    sys.exit(0)

import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd
from dsac_amd import synth
from dsac_amd.shard import ShardRunner
H, W, N = 480, 640, 256
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
frames = {i: synth.chess_like_frame(H, W, seed=1305 + i) for i in range(0, 64, 8)}
perm = torch.from_numpy(synth.fast_permutations(H * W, 8)).to(dev)
run = ShardRunner(eng, st, dev, lambda i: frames[i]["xyz"], 64, 0, 8, N, H, W, frames[0]["cam"], perm, batch=16, emulate=True)
for i in range(120):
    run.step(i)
rows = run.drain()
assert bool(torch.isfinite(rows[run.mine]).all())
run.close()
eng.close()

#!/usr/bin/env python3
"""Round 6: tile / occupancy trades of K2's exact-transform form (k2_variant 84, 89 .. 93) at ONE frame of 256 hypotheses (the latency case) and at the bench
shape (16 frames), against the fp32 forms' auto policy.  us per K2 launch, dispatch-attached events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402

H, W = 480, 640
P = H * W
dev = torch.device("cuda", 0)
# (90 .. 92, <64, 4 x 64> / <64, 2 x 128> with several waves per workgroup, were measured with this script and removed: profiles/r06_k2_exact_tiles.txt)
MODES = [("fp32 forms, auto", 0, -1), ("exact, auto", 1, -1), ("exact <64,256> 1 wave/wg, 2w", 1, 84), ("exact <64,256>, 2w, one-transcendental tail", 1, 85), ("exact <32,256>, 2w", 1, 89),
         ("exact <64, 64> 1 wave/wg, 3w", 1, 93), ("exact <64, 64>, 3w, one-transcendental tail", 1, 94), ("exact <64, 64>, 4w, one-transcendental tail", 1, 95)]


def timed(eng, fn, reps=20):
    for _ in range(3):
        fn()
    eng.synchronize()
    eng.profile_read(0, reset=True)
    for _ in range(reps):
        fn()
    eng.synchronize()
    ms, n = eng.profile_read(0, reset=True)
    return ms / n * 1e3


def main():
    st = torch.cuda.Stream(device=dev)
    eng = dsac_amd.Engine(0, stream=st)
    eng.profile_enable(True, stride=1)
    f64 = dict(dtype=torch.float64, device=dev)
    for F in (1, 16):
        N = 256
        frames = [synth.chess_like_frame(H, W, seed=2305 + f) for f in range(F)]
        xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
        err = torch.empty(F * N, P, dtype=torch.float32, device=dev)
        out = (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev), torch.zeros(F * N, **f64),
               torch.zeros(F * N, **f64), torch.zeros(F, **f64), torch.zeros(F, 6, **f64))
        eng.set_frames(xyz, None, H, W, frames[0]["cam"], borrow=True)
        for _ in range(100 if F > 1 else 400):
            eng.scoreHypothesesFrames(N, seed=1, max_tries=1 << 16, err=err, out=out)
        eng.synchronize()
        for rnd in range(2):
            for name, ex, var in MODES:
                eng.set_option("k2_exact_auto", 1 if ex else 0)
                eng.set_option("k2_flags", (1 << 28) if ex else 0)
                eng.set_option("k2_variant", var)
                us = timed(eng, lambda: eng.scoreHypothesesFrames(N, seed=7, max_tries=1 << 16, err=err, out=out), reps=20 if F == 1 else 10)
                print("%2d frame(s) x 256 x 640x480, err + soft   %-46s %8.1f us   %6.0f GB/s" % (F, name, us, F * (12 * P + 48 * N + 4 * N * P + 4 * N) / us / 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    main()

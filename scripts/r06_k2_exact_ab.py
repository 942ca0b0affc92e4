#!/usr/bin/env python3
"""Round 6: A/B of K2's exact-transform form (k2_flags bit 28, split fp16 records) against the default form and the precise mode (k2_flags bit 25): us per launch at the bench shape (16 frames x 256 hypotheses x 640x480, error images
+ soft-inlier sums) and at BASELINE configs[2] (4096 random poses, error images + sums / error images only), alternating on one box.
Writes profiles-ready text to stdout."""
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
PRECISE = 1 << 25
RECLO = 1 << 27
EXACT = 1 << 28
# (name, k2_flags, k2_variant, k2_exact_auto): since round 6 the auto policy takes the exact form, "fast" switches that off
MODES = [("fast", 0, -1, 0), ("precise", PRECISE, -1, 1), ("reclo<64,256,2w>", RECLO, 81, 1), ("exact (auto policy)", 0, -1, 1), ("exact<64,256,2w>", EXACT, 84, 1),
         ("exact<32,256,2w>", EXACT, 89, 1), ("exact<64,64,3w>", EXACT, 93, 1), ("exactrsq<64,256,2w> one-transcendental tail", EXACT, 85, 1),
         ("exactrsq<64,64,3w> one-transcendental tail", EXACT, 94, 1), ("exactrsq<64,64,4w> one-transcendental tail", EXACT, 95, 1)]
if os.environ.get("DSAC_AB_MODES"):
    MODES = [m for m in MODES if m[0].split("<")[0].split(" ")[0] in os.environ["DSAC_AB_MODES"].split(",")]


def bytes_k2(N, frames=1, err=True):
    return frames * (12 * P + 48 * N + (4 * N * P if err else 0) + 4 * N)


def timed(eng, fn, reps=12):
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
    F, N = 16, 256
    frames = [synth.chess_like_frame(H, W, seed=2305 + f) for f in range(F)]
    xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))).to(dev)
    f64 = dict(dtype=torch.float64, device=dev)
    err = torch.empty(F * N, P, dtype=torch.float32, device=dev)
    err.zero_()
    out = (torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev), torch.zeros(F * N, **f64),
           torch.zeros(F * N, **f64), torch.zeros(F, **f64), torch.zeros(F, 6, **f64))
    torch.cuda.synchronize()
    # settle the clock
    eng.set_frames(xyz, None, H, W, frames[0]["cam"], borrow=True)
    for _ in range(300):
        eng.scoreHypothesesFrames(N, seed=1, max_tries=1 << 16, err=err, out=out)
    eng.synchronize()
    print("# K2 exact-transform form A/B (scripts/r06_k2_exact_ab.py), us per launch, HIP events on the K2 dispatch, alternating on one box")
    rows = []
    for rnd in range(int(os.environ.get('DSAC_AB_ROUNDS', '3'))):
        for name, flags, var, auto in MODES:
            eng.set_option("k2_variant", var)
            eng.set_option("k2_flags", flags)
            eng.set_option("k2_exact_auto", auto)
            us = timed(eng, lambda: eng.scoreHypothesesFrames(N, seed=7, max_tries=1 << 16, err=err, out=out))
            rows.append(("16 x 256 x 640x480, err + soft", name, us, bytes_k2(N, F) / us / 1e3))
    rp = torch.from_numpy(synth.random_poses(4096, seed=7) + np.array([0, 0, 0, 0, 0, 2500.0])).to(dev)
    fr = synth.roofline_frame(H, W, seed=7)
    eng.set_frame(torch.from_numpy(fr["xyz"]).to(dev), None, H, W, fr["cam"], borrow=True)
    soft = torch.zeros(4096, **f64)
    for rnd in range(int(os.environ.get('DSAC_AB_ROUNDS2', '2'))):
        for name, flags, var, auto in MODES:
            eng.set_option("k2_variant", var)
            eng.set_option("k2_flags", flags)
            eng.set_option("k2_exact_auto", auto)
            us = timed(eng, lambda: eng.reproject(rp, N=4096, err=err, soft=soft), reps=8)
            rows.append(("configs[2] N = 4096, err + soft", name, us, bytes_k2(4096) / us / 1e3))
            us = timed(eng, lambda: eng.reproject(rp, N=4096, err=err), reps=8)
            rows.append(("configs[2] N = 4096, err only", name, us, bytes_k2(4096) / us / 1e3))
            us = timed(eng, lambda: eng.reproject(rp, N=4096, soft=soft), reps=8)
            rows.append(("configs[2] N = 4096, soft only (no stores)", name, us, float("nan")))
    eng.set_option("k2_flags", 0)
    eng.set_option("k2_variant", -1)
    eng.set_option("k2_exact_auto", 1)
    for r in rows:
        print("%-44s %-18s %8.1f us   %6.0f GB/s (algorithmic bytes)" % r)
    eng.close()


if __name__ == "__main__":
    main()

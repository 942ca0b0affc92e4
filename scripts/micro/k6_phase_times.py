"""Where one refinement's time goes: an instrumented build of K6 (build/ab/libdsac_hip_k6timing.so through DSAC_HIP_LIB; /tmp recipe in profiles/r05_k6_phases.txt)
returns 100 MHz ticks per phase instead of the refined pose -- waiting for a continuation batch's loads, the four fp64 residuals per lane, the in-order
compaction, everything from the end of the walk to the end of the step (head prefetch + LM solve), the first batch's wait, and the kernel's total."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
from dsac_amd.capi import lib, ptr, check
dev = torch.device("cuda:0")
eng = dsac_amd.Engine(0)
rng = np.random.default_rng(5)
for (H, W, outl, B, spread, what) in ((480, 640, 0.3, 1, 0.0, "one problem, good pose"), (40, 40, 0.3, 1, 0.0, "one problem, good pose"),
                                      (480, 640, 0.9, 1, 0.0, "one problem, 10 % inliers"), (40, 40, 0.9, 1, 0.0, "one problem, 10 % inliers"),
                                      (480, 640, 0.3, 128, 1.0, "128 problems, random poses"), (40, 40, 0.3, 256, 1.0, "256 problems, random poses")):
    P = H * W
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40), outlier_frac=outl)
    xyz = torch.from_numpy(fr["xyz"]).to(dev)
    uv = torch.from_numpy(fr["uv"]).to(dev) if H == 40 else None
    eng.set_frame(xyz, uv, H, W, fr["cam"], borrow=True)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    init = fr["gt_pose"][None, :] + rng.normal(size=(B, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0]) * (1.0 + 30.0 * spread)
    init_d = torch.from_numpy(np.ascontiguousarray(init)).to(dev)
    out = torch.zeros(B, 6, dtype=torch.float64, device=dev)
    sd = torch.zeros(B, dtype=torch.int32, device=dev)
    for i in range(3):
        check(eng._ctx, lib.dsac_refine(eng._ctx, B, ptr(init_d), ptr(perm), 8, 100, 50, 10.0, None, None, ptr(out), None, ptr(sd)))
    eng.synchronize()
    t = out.cpu().numpy() * 0.01  # ticks of 10 ns -> us
    m = t.mean(0)
    print("K6 %3dx%-3d %-28s: total %8.1f us = continuation loads %7.1f + residuals %7.1f + compaction %6.1f + prefetch / LM %6.1f + first-batch wait %5.1f (+ %5.1f other); steps %.2f" %
          (W, H, what, m[5], m[0], m[1], m[2], m[3], m[4], m[5] - m[:5].sum(), sd.float().mean().item()), flush=True)

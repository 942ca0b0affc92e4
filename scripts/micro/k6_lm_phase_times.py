"""Inside the Levenberg-Marquardt chain of one refinement: an instrumented K6 (scripts/micro/k6_lm_timing.patch -> build/ab/libdsac_hip_k6timing2.so through
DSAC_HIP_LIB) returns 100 MHz ticks for Rodrigues (+ its derivative), the per-correspondence pass, the wave-wide reduction, the 6x6 solves, and the counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
from dsac_amd.capi import lib, ptr, check
dev = torch.device("cuda:0")
eng = dsac_amd.Engine(0)
rng = np.random.default_rng(5)
for (H, W, outl) in ((480, 640, 0.3), (40, 40, 0.3), (40, 40, 0.9)):
    P = H * W
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40), outlier_frac=outl)
    xyz = torch.from_numpy(fr["xyz"]).to(dev)
    uv = torch.from_numpy(fr["uv"]).to(dev) if H == 40 else None
    eng.set_frame(xyz, uv, H, W, fr["cam"], borrow=True)
    perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    init = fr["gt_pose"][None, :] + rng.normal(size=(1, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0])
    init_d = torch.from_numpy(np.ascontiguousarray(init)).to(dev)
    out = torch.zeros(1, 6, dtype=torch.float64, device=dev)
    sd = torch.zeros(1, dtype=torch.int32, device=dev)
    for i in range(3):
        check(eng._ctx, lib.dsac_refine(eng._ctx, 1, ptr(init_d), ptr(perm), 8, 100, 50, 10.0, None, None, ptr(out), None, ptr(sd)))
    eng.synchronize()
    t = out.cpu().numpy()[0]
    solves = int(t[4] // 1e6); evJ = int((t[4] % 1e6) // 1000); ev0 = int(t[4] % 1000)
    print("K6 %3dx%-3d outliers %.0f %%: total %6.1f us; LM: Rodrigues + dR %5.1f us, correspondences %5.1f us, reductions %5.1f us, 6x6 solves %5.1f us; %d evaluations with normal equations, %d without, %d solves over %d steps"
          % (W, H, 100 * outl, t[5] * 0.01, t[0] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, evJ, ev0, solves, int(sd.item())), flush=True)

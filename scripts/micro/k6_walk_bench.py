"""K6 (k_refine) where the inlier walk is long: one problem on a good pose (the walk stops in its first batch), one problem on a hard frame (10 % inliers: several
batches per step), and many problems from random poses (the DSAC variant refines every hypothesis: most walk the whole map).  Device pointers, torch events on the
engine's stream; a checksum of the refined poses so that two builds can be compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
from dsac_amd.capi import lib, ptr, check
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
rng = np.random.default_rng(5)
for (H, W, outl, B, spread, what) in ((480, 640, 0.3, 1, 0.0, "one problem, good pose"), (40, 40, 0.3, 1, 0.0, "one problem, good pose"),
                                      (480, 640, 0.9, 1, 0.0, "one problem, 10 % inliers"), (40, 40, 0.9, 1, 0.0, "one problem, 10 % inliers"),
                                      (480, 640, 0.95, 16, 0.0, "16 problems, 5 % inliers"),
                                      (40, 40, 0.3, 256, 1.0, "256 problems, random poses"), (480, 640, 0.3, 128, 1.0, "128 problems, random poses"),
                                      (40, 40, 0.3, 4096, 1.0, "4096 problems, random poses")):
    if os.environ.get("DSAC_K6_CASE") and os.environ["DSAC_K6_CASE"] not in what: continue
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
    def call():
        check(eng._ctx, lib.dsac_refine(eng._ctx, B, ptr(init_d), ptr(perm), 8, 100, 50, 10.0, None, None, ptr(out), None, ptr(sd)))
    for waves in [int(v) for v in os.environ.get("DSAC_K6_WAVES", "0").split(",")]:  # round 6: waves per problem of the walk ("k6_waves": 0 = auto, 1, 2, 4, 8)
        eng.set_option("k6_waves", abs(waves) % 100)
        eng.set_option("k6_walk_exact", 1 if waves == -100 else 0)
        if os.environ.get("DSAC_K6_SCAN_TUNE"): eng.set_option("k6_scan_tune", int(os.environ["DSAC_K6_SCAN_TUNE"], 0))  # -100: auto with the walk's fp32 filter switched off (A/B)
        for i in range(3): call()
        eng.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if B * P < 5e7 else 5
        with torch.cuda.stream(st):
            a.record(st)
            for i in range(reps): call()
            b.record(st)
        eng.synchronize(); torch.cuda.synchronize()
        o = out.cpu().numpy()
        print("K6 %3dx%-3d %-28s k6_waves %d: %9.1f us per launch; steps done mean %.2f; checksum %.17e" % (W, H, what, waves, a.elapsed_time(b) * 1e3 / reps, sd.float().mean().item(), float(np.abs(o).sum())), flush=True)

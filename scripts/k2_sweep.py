"""K2 micro-sweep on the GPU: kernel variants x outputs x N, timed with the engine's HIP-event hooks."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth

dev = torch.device("cuda:0")
H, W = 480, 640
P = H * W
fr = synth.roofline_frame(H, W)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
# pure-write ceiling: torch fill of the same buffer size
for N in (256, 4096):
    buf = torch.empty(N, P, dtype=torch.float32, device=dev)
    for _ in range(3): buf.fill_(1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): buf.fill_(1.0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("fill N=%d: %.1f us  %.0f GB/s" % (N, ms * 1e3, 4.0 * N * P / ms / 1e6))
    del buf
res = []
for variant, order, kf in ((4, 0, 0), (4, 0, 1), (5, 0, 0), (0, 1, 0)):
    os.environ["DSAC_K2_VARIANT"] = str(variant)
    os.environ["DSAC_K2_ORDER"] = str(order)
    os.environ["DSAC_K2_FLAGS"] = str(kf)
    eng = dsac_amd.Engine(0)
    eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
    eng.profile_enable(True)
    for N in (256, 4096):
        poses = torch.from_numpy(synth.random_poses(N, seed=7) + np.array([0, 0, 0, 0, 0, 2500.0])).to(dev)
        err = torch.empty(N, P, dtype=torch.float32, device=dev)
        soft = torch.zeros(N, dtype=torch.float64, device=dev)
        for mode in ("err", "soft", "both"):
            kw = dict(err=err if mode != "soft" else None, soft=soft if mode != "err" else None)
            for _ in range(3): eng.reproject(poses, N=N, **kw)
            eng.synchronize(); eng.profile_read(0)
            reps = 20 if N == 256 else 8
            for _ in range(reps): eng.reproject(poses, N=N, **kw)
            eng.synchronize()
            ms, n = eng.profile_read(0)
            us = ms / n * 1e3
            ab = 12 * P + 48 * N + (4 * N * P if mode != "soft" else 0) + 4 * N
            print("variant %d order %d flags %d N=%4d %-4s: %8.1f us  %7.0f GB/s (alg)  %.2f Gpair/s" % (variant, order, kf, N, mode, us, ab / us / 1e3, N * P / us / 1e3))
            res.append(dict(variant=variant, N=N, mode=mode, us=us, gbs=ab / us / 1e3))
        del err
    eng.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/k2_sweep.json", "w"))

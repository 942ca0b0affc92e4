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
import collections
acc = collections.defaultdict(list)
configs = ((6, 1, 0), (4, 1, 0), (4, 0, 0), (0, 1, 0))
for rep in range(3):
  for variant, order, kf in configs:
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
        for mode in ("err", "both"):
            kw = dict(err=err if mode != "soft" else None, soft=soft if mode != "err" else None)
            for _ in range(3): eng.reproject(poses, N=N, **kw)
            eng.synchronize(); eng.profile_read(0)
            reps = 30 if N == 256 else 8
            for _ in range(reps): eng.reproject(poses, N=N, **kw)
            eng.synchronize()
            ms, n = eng.profile_read(0)
            acc[(variant, order, kf, N, mode)].append(ms / n * 1e3)
        del err
    eng.close()
for (variant, order, kf, N, mode), v in sorted(acc.items(), key=lambda kv: (kv[0][3], kv[0][4], kv[0][0])):
    us = min(v)
    ab = 12 * P + 48 * N + (4 * N * P if mode != "soft" else 0) + 4 * N
    print("variant %d order %d flags %d N=%4d %-4s: best %8.1f us (runs %s)  %7.0f GB/s (alg)" % (variant, order, kf, N, mode, us, " ".join("%.0f" % x for x in v), ab / us / 1e3))
    res.append(dict(variant=variant, order=order, N=N, mode=mode, us=us, gbs=ab / us / 1e3))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/k2_sweep.json", "w"))

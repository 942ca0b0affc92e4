"""K4 (score backward) timing on the GPU with the engine's HIP-event hooks; algorithmic bytes per SURVEY 8(d)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth

dev = torch.device("cuda:0")
H, W = 480, 640
P = H * W
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
eng = dsac_amd.Engine(0)
eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
eng.profile_enable(True)
res = []
for N in (256, 1024):
    poses = torch.zeros(N, 6, dtype=torch.float64, device=dev)
    sets = torch.zeros(N, 4, dtype=torch.int32, device=dev)
    ok = torch.zeros(N, dtype=torch.uint8, device=dev)
    eng.sample(N, seed=7, out=(poses, sets, ok))
    d_err = torch.randn(N, P, dtype=torch.float32, device=dev) * 1e-3
    g = torch.randn(N, dtype=torch.float64, device=dev)
    grad = torch.zeros(P, 3, dtype=torch.float64, device=dev)
    dpnp = torch.zeros(N, 72, dtype=torch.float64, device=dev)
    eng.dPNP(sets, out=dpnp)
    for mode in ("d_err", "soft"):
        def run():
            if mode == "d_err":
                eng.dScore(poses, sets, d_err, dpnp=dpnp, grad=grad)
            else:
                eng.dSoftScore(poses, sets, g, dpnp=dpnp, grad=grad)
        # settle the clocks first (as bench.py's prewarm does for K2): ten launches of 0.1-0.5 ms straight out of an idle GPU are timed at the idle
        # clock -- the same binary reads 112 us cold and 102 us after 50 ms of work (in the trainer K4 follows the score CNN's backward: warm)
        import time
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.25:
            for _ in range(20): run()
            eng.synchronize()
        eng.profile_read(1)
        # the whole K4 stage (prep + main pass + grad reduce + support scatter) on the engine's stream: device pointers only, so the
        # calls just enqueue; wall time over 40 back-to-back calls
        t0 = time.perf_counter()
        for _ in range(40): run()
        eng.synchronize()
        stage_us = (time.perf_counter() - t0) / 40 * 1e6
        ms, n = eng.profile_read(1)
        us = ms / n * 1e3
        ab = (4 * N * P if mode == "d_err" else 0) + 12 * P + 48 * N + 48 * N + 12 * P
        print("K4 N=%4d %-5s: main pass %8.1f us  %7.0f GB/s (alg)  %.2f Gpair/s | whole stage %8.1f us  %7.0f GB/s (alg, frac %.3f of 8 TB/s)" %
              (N, mode, us, ab / us / 1e3, N * P / us / 1e3, stage_us, ab / stage_us / 1e3, ab / stage_us / 1e3 / 8000))
        res.append(dict(N=N, mode=mode, us=us, gbs=ab / us / 1e3, stage_us=stage_us, stage_gbs=ab / stage_us / 1e3))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/k4_bench.json", "w"))

"""K4 kernel forms side by side in one process (round 3): main pass (HIP events of the engine) and whole stage, N = 256 / 1024 x 640x480.
usage: python scripts/r03_k4_sweep.py [variant ...]   (k4_variant values: form + 10 * tile code + 100 * workgroups per CU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsac_amd
from dsac_amd import synth

variants = [int(v) for v in sys.argv[1:]] or [-1, 2, 6, 7]
dev = torch.device("cuda:0")
H, W = 480, 640
P = H * W
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
eng = dsac_amd.Engine(0)
eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
eng.profile_enable(True)
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
    ref = {}
    for rep in range(2):
        for v in variants:
            eng.set_option("k4_variant", v)
            for mode in ("d_err", "soft"):
                def run():
                    if mode == "d_err":
                        eng.dScore(poses, sets, d_err, dpnp=dpnp, grad=grad)
                    else:
                        eng.dSoftScore(poses, sets, g, dpnp=dpnp, grad=grad)
                grad.zero_()
                run()
                eng.synchronize()
                key = (mode,)
                if key not in ref:
                    ref[key] = grad.clone()
                dmax = float((grad - ref[key]).abs().max() / ref[key].abs().max())
                for _ in range(3):
                    run()
                eng.synchronize(); eng.profile_read(1)
                t0 = time.perf_counter()
                for _ in range(10):
                    run()
                eng.synchronize()
                stage_us = (time.perf_counter() - t0) / 10 * 1e6
                ms, n = eng.profile_read(1)
                us = ms / n * 1e3
                ab = (4 * N * P if mode == "d_err" else 0) + 12 * P + 48 * N + 48 * N + 12 * P
                print("K4 N=%4d k4_variant %4d %-5s: main pass %7.1f us %6.0f GB/s (%.3f of 8 TB/s) | stage %7.1f us | vs first form %.1e" %
                      (N, v, mode, us, ab / us / 1e3, ab / us / 1e3 / 8000, stage_us, dmax), flush=True)
    del d_err

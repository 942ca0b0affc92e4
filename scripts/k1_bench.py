"""K1 (sample + P3P) alone, timed with torch events on the engine's stream: N hypotheses on one 640x480 frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
dev = torch.device("cuda:0")
fr = synth.chess_like_frame(480, 640, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
eng.set_frame(xyz, None, 480, 640, fr["cam"], borrow=True)
for N in (256, 512, 1024, 2048, 4096):
    poses = torch.zeros(N, 6, dtype=torch.float64, device=dev); sets = torch.zeros(N, 4, dtype=torch.int32, device=dev); ok = torch.zeros(N, dtype=torch.uint8, device=dev)
    for i in range(3): eng.sample(N, seed=i, out=(poses, sets, ok))
    eng.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        a.record(st)
        for i in range(20): eng.sample(N, seed=100 + i, out=(poses, sets, ok))
        b.record(st)
    eng.synchronize(); torch.cuda.synchronize()
    print("K1 N=%5d: %7.1f us per launch  (%.1f ns per hypothesis)  ok %.3f" % (N, a.elapsed_time(b) * 1e3 / 20, a.elapsed_time(b) * 1e6 / 20 / N, ok.float().mean().item()))

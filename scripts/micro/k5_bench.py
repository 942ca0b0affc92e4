"""K5 (dPNP: 24 P3P solves per minimal set) alone, timed with torch events on the engine's stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
dev = torch.device("cuda:0")
fr = synth.chess_like_frame(480, 640, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
eng.set_frame(xyz, None, 480, 640, fr["cam"], borrow=True)
for N in (256, 1024, 4096):
    poses = torch.zeros(N, 6, dtype=torch.float64, device=dev); sets = torch.zeros(N, 4, dtype=torch.int32, device=dev); ok = torch.zeros(N, dtype=torch.uint8, device=dev)
    eng.sample(N, seed=3, out=(poses, sets, ok))
    J = torch.zeros(N, 6, 12, dtype=torch.float64, device=dev)
    for i in range(3): eng.dPNP(sets, out=J)
    eng.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        a.record(st)
        for i in range(20): eng.dPNP(sets, out=J)
        b.record(st)
    eng.synchronize(); torch.cuda.synchronize()
    print("K5 N=%5d: %7.1f us per launch  (%.1f ns per minimal set)  |J| %.6e" % (N, a.elapsed_time(b) * 1e3 / 20, a.elapsed_time(b) * 1e6 / 20 / N, J.abs().sum().item()))

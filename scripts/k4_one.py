"""One K4 configuration (for rocprofv3 PMC passes): python scripts/k4_one.py N mode(d_err|soft) reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsac_amd
from dsac_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "d_err"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
H, W = 480, 640
P = H * W
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
eng = dsac_amd.Engine(0)
eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
poses = torch.zeros(N, 6, dtype=torch.float64, device=dev); sets = torch.zeros(N, 4, dtype=torch.int32, device=dev); ok = torch.zeros(N, dtype=torch.uint8, device=dev)
eng.sample(N, seed=7, out=(poses, sets, ok))
d_err = torch.randn(N, P, dtype=torch.float32, device=dev) * 1e-3
g = torch.randn(N, dtype=torch.float64, device=dev)
grad = torch.zeros(P, 3, dtype=torch.float64, device=dev)
dpnp = torch.zeros(N, 72, dtype=torch.float64, device=dev)
eng.dPNP(sets, out=dpnp)
for _ in range(reps):
    if mode == "d_err":
        eng.dScore(poses, sets, d_err, dpnp=dpnp, grad=grad)
    else:
        eng.dSoftScore(poses, sets, g, dpnp=dpnp, grad=grad)
eng.synchronize()

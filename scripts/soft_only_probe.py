"""Scoring WITHOUT the error-image side output: K1 -> K2 (soft-inlier sums only) -> K3 on the bench's 8-frame batch and on one frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
dev = torch.device("cuda:0")
eng = dsac_amd.Engine(0)
N, H, W = 256, 480, 640
for nf in (8, 1):
    xs = torch.from_numpy(np.ascontiguousarray(np.stack([synth.chess_like_frame(H, W, seed=1305 + k)["xyz"] for k in range(nf)]))).to(dev)
    cam = synth.chess_like_frame(40, 40)["cam"]
    if nf == 1: eng.set_frame(xs[0], None, H, W, cam, borrow=True)
    else: eng.set_frames(xs, None, H, W, cam, borrow=True)
    nn = N * nf
    f64 = dict(dtype=torch.float64, device=dev)
    o = (torch.zeros(nn, 6, **f64), torch.zeros(nn, 4, dtype=torch.int32, device=dev), torch.zeros(nn, dtype=torch.uint8, device=dev), torch.zeros(nn, **f64),
         torch.zeros(nn, **f64), torch.zeros(nf, **f64), torch.zeros(nf, 6, **f64))
    eng.profile_enable(True, stride=1)
    def one(i):
        if nf == 1: eng.scoreHypotheses(N, seed=100 + i, scale=0.1, err=None, out=(o[0], o[1], o[2], o[3], o[4], o[5][:1], o[6][0]))
        else: eng.scoreHypothesesFrames(N, seed=100 + i, scale=0.1, err=None, out=o)
    for i in range(20): one(i)
    eng.synchronize(); eng.profile_read(0, reset=True)
    n = 200
    t0 = time.perf_counter()
    for i in range(n): one(20 + i)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / n
    ms, c = eng.profile_read(0, reset=True)
    print("soft-inlier scoring without error images, %d frame(s) x %d hypotheses x %dx%d: %.1f us per step, %.2f M hyp/s, K2 %.1f us per launch" %
          (nf, N, W, H, dt * 1e6, nn / dt / 1e6, ms / max(1, c) * 1e3))

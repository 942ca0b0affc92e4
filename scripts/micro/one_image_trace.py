"""One 640x480 image per call through dsac_process_images, 200 calls back to back on one stream (the `process_image.640x480` leg of bench.py): run under
rocprofv3 --kernel-trace --stats for the per-kernel breakdown (scripts/micro/one_image_trace.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsac_amd
from dsac_amd import synth
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
if os.environ.get("DSAC_K2_EXACT_AUTO_OFF"): eng.set_option("k2_exact_auto", 0)
H, W, N = 480, 640, 256
P = H * W
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(fr["xyz"]).to(dev)
eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)
perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
f64 = dict(dtype=torch.float64, device=dev)
err = torch.empty(N, P, dtype=torch.float32, device=dev)
out = dict(hyps=torch.zeros(N, 6, **f64), sampledPoints=torch.zeros(N, 4, dtype=torch.int32, device=dev), ok=torch.zeros(N, dtype=torch.uint8, device=dev), scores=torch.zeros(N, **f64),
           sfScores=torch.zeros(N, **f64), sfEntropy=torch.zeros(1, **f64), avgHyp=torch.zeros(1, 6, **f64), refAvgHyp=torch.zeros(1, 6, **f64), refSteps=torch.zeros(1, dtype=torch.int32, device=dev),
           out4=torch.zeros(1, 4, **f64))
gt = torch.zeros(1, 6, **f64)
def proc(i):
    eng.processImages(N, perm, gt_jp6=gt, seed=1000 + i, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=err, out=out)
for i in range(20): proc(i)
eng.synchronize()
n = 200
t = time.perf_counter()
for i in range(n): proc(20 + i)
eng.synchronize()
print("one image per call, %d calls back to back: %.1f us per image" % (n, (time.perf_counter() - t) / n * 1e6), flush=True)

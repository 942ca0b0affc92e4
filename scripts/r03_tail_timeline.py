"""Kernel timeline of dsac_process_images with the refinement tail deferred (dsac_set_option "pi_defer_tail"): run under
rocprofv3 --kernel-trace, then `python scripts/r03_tail_timeline.py <kernel_trace.csv>` prints how much of every K6 launch ran while a K1 / K2 of the
next batch was executing."""
import sys, csv
if len(sys.argv) > 1:
    rows = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
    k6 = [(s, e) for n, s, e in rows if "k_refine" in n]
    big = [(s, e) for n, s, e in rows if "k_reproject" in n or "k_sample" in n]
    tot = ov = 0
    for s, e in k6[-40:]:
        tot += e - s
        ov += sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in big)
    print("last %d K6 launches: mean %.1f us each, %.0f %% of their time under a K1 / K2 launch" % (len(k6[-40:]), tot / max(1, len(k6[-40:])) / 1e3, 100.0 * ov / max(1, tot)))
    sys.exit(0)
import os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd
from dsac_amd import synth
H, W, F, N = 480, 640, 16, 256
P = H * W
dev = torch.device("cuda", 0)
eng = dsac_amd.Engine(0)
fr = synth.chess_like_frame(H, W, seed=1305)
xyz = torch.from_numpy(np.ascontiguousarray(np.stack([fr["xyz"]] * F))).to(dev)
perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
gts = torch.zeros(F, 6, dtype=torch.float64, device=dev)
n = F * N
o = dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev), ok=torch.zeros(n, dtype=torch.uint8, device=dev),
         scores=torch.zeros(n, dtype=torch.float64, device=dev), sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(F, dtype=torch.float64, device=dev),
         avgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev),
         refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, dtype=torch.float64, device=dev))
err = torch.empty(n, P, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
eng.set_frames(xyz, None, H, W, fr["cam"], borrow=True)
eng.set_option("pi_defer_tail", 1)
for i in range(60):
    eng.processImages(N, perm, gt_jp6=gts, seed=5 + i, err=err, out=o)
eng.synchronize()

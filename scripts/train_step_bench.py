"""one end-to-end training step per rank (one frame per GPU), reference CNN architectures with
random weights, geometry on the HIP engine, gradients of both CNNs (~157 MB fp32) all-reduced over RCCL.

  python scripts/train_step_bench.py [--steps 20] [--warmup 3]                       # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_step_bench.py
Prints one JSON line on rank 0 (frames/s over all ranks, ms per step and its split)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from dsac_amd import dist as ddist, e2e, synth

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--hyps", type=int, default=256)
ap.add_argument("--sub-sample", type=float, default=0.01)
a = ap.parse_args()
rank, world, local = ddist.init(backend=os.environ.get("DSAC_BENCH_BACKEND"))
dev_index = local if torch.cuda.device_count() > local else 0
ts = e2e.TrainStep(dev_index, hyps=a.hyps, sub_sample=a.sub_sample)
dev = ts.dev
fr = synth.chess_like_frame(40, 40, seed=1305 + rank, quantise_int16=True)
patches = torch.rand(1600, 3, 42, 42, device=dev) * 255
uv = torch.as_tensor(fr["uv"], device=dev)
off = torch.as_tensor(fr["xyz"], device=dev)
perm = synth.fast_permutations(1600, 8)
# ground truth in the jp convention, a little off the rendering pose so that the loss has a gradient
from dsac_amd.synth import rodrigues
Rcv = rodrigues(fr["gt_pose"][:3] + np.array([0.01, -0.02, 0.01])); tcv = fr["gt_pose"][3:] + np.array([5.0, -8.0, 12.0])
F = np.diag([1.0, -1.0, -1.0]); Rj = F @ Rcv; tj = F @ tcv
th = np.arccos(np.clip((np.trace(Rj) - 1) / 2, -1, 1)); ax = np.array([Rj[2, 1] - Rj[1, 2], Rj[0, 2] - Rj[2, 0], Rj[1, 0] - Rj[0, 1]])
gt = np.concatenate([ax / (2 * np.sin(th)) * th, tj])


def barrier():
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


for i in range(a.warmup):
    ts.step(patches, uv, gt, perm, seed=1000 + i, xyz_offset_mm=off)
barrier()
t0 = time.perf_counter()
tc = 0.0
for i in range(a.steps):
    out = ts.step(patches, uv, gt, perm, seed=2000 + i, xyz_offset_mm=off)
barrier()
dt = time.perf_counter() - t0
t = torch.tensor([dt], dtype=torch.float64, device=dev)
if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    nparam = sum(p.numel() for p in ts.params())
    print(json.dumps(dict(metric="training frames / s (end-to-end step: 2 CNNs + geometry fwd/bwd + gradient all-reduce)", value=world * a.steps / float(t.item()),
                          unit="frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * float(t.item()) / a.steps, scaling="weak",
                          data="synthetic", config=dict(workload="config 5: 1 frame/GPU, 40x40 map, %d hypotheses" % a.hyps, grad_bytes=4 * nparam,
                                                        collectives_per_step=out["collectives"], last_loss=out["loss"]))))
if world > 1:
    dist.destroy_process_group()

#!/usr/bin/env python3
"""Round 4: where does the step of ONE rank of 8 (8 images x 256 hypotheses x 640x480 per step, BASELINE configs[3]) spend its time?
The same 8-frame batch through four drivers, alternating on one box, K2 timed by its dispatch-attached events, the step by the wall clock over
`reps` steps without host synchronisation:
  A  dsac_score_hypotheses_frames only (K1, K2, reduce, K3)                       -- the floor: no refinement at all
  B  dsac_process_images, tail in stream order (K6 / K7 exposed)
  C  dsac_process_images, tail deferred ("pi_defer_tail"), one joinTail at the end
  D  dsac_amd.shard.ShardRunner.step (C + exchange buffers + one-step-late gather on a side stream), emulated rank 0 of 8
  E  D with the Python-side bound-argument fast path replaced by Engine.processImages (what the marshalling costs)
usage: r04_rank_step_lab.py [frames=8] [reps=60] [rounds=4]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402
from dsac_amd.shard import ShardRunner  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N, H, W = 256, 480, 640
P = H * W
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
eng = dsac_amd.Engine(0, stream=st)
frames = [synth.chess_like_frame(H, W, seed=1305 + i) for i in range(64)]
cam = frames[0]["cam"]
xyz = torch.from_numpy(np.ascontiguousarray(np.stack([frames[i * (64 // F) % 64]["xyz"] for i in range(F)]))).to(dev)
perm = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
gts = torch.zeros(F, 6, dtype=torch.float64, device=dev)
n = F * N
b = dict(hyps=torch.zeros(n, 6, dtype=torch.float64, device=dev), sampledPoints=torch.zeros(n, 4, dtype=torch.int32, device=dev),
         ok=torch.zeros(n, dtype=torch.uint8, device=dev), scores=torch.zeros(n, dtype=torch.float64, device=dev),
         sfScores=torch.zeros(n, dtype=torch.float64, device=dev), sfEntropy=torch.zeros(F, dtype=torch.float64, device=dev),
         avgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev), refAvgHyp=torch.zeros(F, 6, dtype=torch.float64, device=dev),
         refSteps=torch.zeros(F, dtype=torch.int32, device=dev), out4=torch.zeros(F, 4, dtype=torch.float64, device=dev))
err = torch.zeros(n, P, dtype=torch.float32, device=dev)
err_alt = None  # A2 / C2: the same drivers writing the runner's error-image buffer (is a difference in K2 the buffer or the schedule?)
eng.profile_enable(True, stride=1)
runner = ShardRunner(eng, st, dev, lambda i: frames[i]["xyz"], 64, 0, 64 // F, N, H, W, cam, perm, batch=F, emulate=True)
runner.close()
err_alt = runner.err


def drive(mode, reps):
    eng.set_option("device_args", 0)
    eng.set_option("seed_stride", 1)
    eng.set_option("pi_defer_tail", 1 if mode in "Cc" else 0)
    if mode in "DE":
        for key, v in (("device_args", 1 if mode == "D" else 0), ("seed_stride", runner.world), ("pi_defer_tail", 1)):
            eng.set_option(key, v)
    else:
        eng.set_frames(xyz, None, H, W, cam, borrow=True)
    torch.cuda.synchronize(dev)
    eng.profile_read(0, reset=True)
    t0 = time.perf_counter()
    for i in range(reps):
        if mode in "Aa":
            eng.scoreHypothesesFrames(N, seed=1305 + i, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=err if mode == "A" else err_alt,
                                      out=(b["hyps"], b["sampledPoints"], b["ok"], b["scores"], b["sfScores"], b["sfEntropy"], b["avgHyp"]))
        elif mode in "BCc":
            eng.processImages(N, perm, gt_jp6=gts, seed=1305 + i, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=err if mode != "c" else err_alt, out=b)
        elif mode == "D":
            runner.step(i)
        else:  # E: the runner's schedule with the generic Python marshalling
            k = runner.steps_done & 1
            if runner.steps_done >= 1:
                runner._consume(k)
                runner._launch_gather(1 - k)
            if runner.consumed_valid[k]:
                st.wait_event(runner.consumed[k])
            ref_v, out4_v, w_v = runner.ex.views(k)
            eng.set_frames(runner.xyz[0], None, H, W, cam, borrow=True)
            s = runner.scratch[0]
            eng.processImages(N, perm, gt_jp6=runner.gt[:F], seed=1305 + 64 * i, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1,
                              err=runner.err, out=dict(hyps=s["poses"], sampledPoints=s["sets"], ok=s["ok"], scores=s["soft"], sfScores=w_v[:F].view(-1),
                                                       sfEntropy=s["ent"], avgHyp=s["avg"], refAvgHyp=ref_v[:F], refSteps=s["sd"], out4=out4_v[:F]))
            runner._last_slot = k
            runner.steps_done += 1
    host = time.perf_counter() - t0
    if mode in "DE":
        runner.drain()
    eng.joinTail()
    eng.synchronize()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms, cnt = eng.profile_read(0, reset=True)
    return wall / reps * 1e6, host / reps * 1e6, ms / max(1, cnt) * 1e3


MODES = "AaBCcDE"
if os.environ.get("LAB_MAIN_FIRST"):
    # what bench.py --workload config3 --emulate-world 8 does before the emulation: the whole 64-image job (4 batches of 16 per step) on the same engine
    main = ShardRunner(eng, st, dev, lambda i: frames[i]["xyz"], 64, 0, 1, N, H, W, cam, perm, batch=16)
    t0 = time.perf_counter()
    for i in range(25):
        main.step(i)
    main.drain()
    print("main runner: %.1f us per 64-image step" % ((time.perf_counter() - t0) / 25 * 1e6))
    if os.environ.get("LAB_MAIN_FIRST") == "close":
        main.close()
        del main
        torch.cuda.empty_cache()
    runner2 = ShardRunner(eng, st, dev, lambda i: frames[i]["xyz"], 64, 0, 64 // F, N, H, W, cam, perm, batch=F, emulate=True)
    runner2.close()
    runner = runner2
res = {m: [] for m in MODES}
drive("A", 30)
for r in range(rounds):
    for m in (MODES if r % 2 == 0 else MODES[::-1]):
        res[m].append(drive(m, reps))
print("%d frames x %d hypotheses x %dx%d per step, %d steps per measurement, %d rounds: us per step (wall) | host enqueue us per step | K2 us per launch  [medians; min-max of the step]" % (F, N, W, H, reps, rounds))
names = dict(a="A2 as A, error images into the runner's buffer", c="C2 as C, error images into the runner's buffer", A="A  score_hypotheses_frames (K1 K2 K3 only)", B="B  process_images, tail in stream order", C="C  process_images, tail deferred",
             D="D  ShardRunner.step (bound arguments)", E="E  ShardRunner schedule, generic marshalling")
for m in MODES:
    a = np.array(res[m])
    print("%-48s %8.1f | %8.1f | %8.1f   [%0.1f - %0.1f]" % (names[m], np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), a[:, 0].min(), a[:, 0].max()))
eng.close()

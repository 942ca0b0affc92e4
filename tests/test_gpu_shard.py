"""dsac_amd.shard.ShardRunner -- BASELINE.json configs[3] for one rank: batches through dsac_process_images with the refinement tail deferred across batch
AND step boundaries, results written straight into the exchange buffer, the gather of a step launched at the top of the next step on a side stream
(dsac_tail_wait) and consumed one step later, no host synchronisation inside a step.  Parity: every image's row equals the in-order, per-image,
nothing-deferred call with the same seed bit for bit -- for the unsharded run and for every emulated rank of a 3-rank sharding (whose union is the
unsharded result: "seed_stride" keeps an image's seed independent of the sharding)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W, N, NIMG, STEPS = 48, 64, 128, 12, 3


def _reference_rows(synth, perm_np, gts, step):
    """The per-image loop of core/test_ransac_softam.cpp:97-157, one image per call, in stream order (no deferral), host results."""
    import dsac_amd
    rows = np.zeros((NIMG, 10 + N))
    with dsac_amd.Engine(0) as eng:
        for i in range(NIMG):
            fr = synth.chess_like_frame(H, W, seed=700 + i)
            eng.set_frame(fr["xyz"], None, H, W, fr["cam"])
            r = eng.processImages(N, perm_np, gt_jp6=gts[i:i + 1], seed=1305 + 64 * step + i, max_tries=1 << 16)
            rows[i, :6], rows[i, 6:10], rows[i, 10:] = r["refAvgHyp"][0], r["out4"][0], r["sfScores"]
    return rows


@pytest.mark.parametrize("world,defer", [(1, 2), (3, 2), (3, 1), (1, 0)])
def test_shard_runner_equals_the_in_order_per_image_loop(synth, orc, world, defer):
    import torch
    import dsac_amd
    from dsac_amd.shard import ShardRunner
    dev = torch.device("cuda", 0)
    perm_np = synth.fast_permutations(H * W, 8)
    perm = torch.from_numpy(perm_np).to(dev)
    frames = [synth.chess_like_frame(H, W, seed=700 + i) for i in range(NIMG)]
    gts = np.stack([orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for fr in frames])
    want = _reference_rows(synth, perm_np, gts, STEPS - 1)
    got = np.zeros_like(want)
    for rank in range(world):
        st = torch.cuda.Stream(device=dev)
        eng = dsac_amd.Engine(0, stream=st)
        try:
            run = ShardRunner(eng, st, dev, lambda i: frames[i]["xyz"], NIMG, rank, world, N, H, W, frames[0]["cam"], perm, gt_of=lambda i: gts[i], batch=4,
                              emulate=world > 1, defer=defer)  # 2: score + refinement tail of a batch under the next one, 1: refinement only, 0: in order
            assert run.mine == list(range(rank, NIMG, world))
            for s in range(STEPS):
                run.step(s)      # never waits: the tail of step s runs under step s + 1, the gather of step s is launched in step s + 1
            rows = run.drain().numpy()
            got[run.mine] = rows[run.mine]
            other = [i for i in range(NIMG) if i not in run.mine]
            assert not rows[other].any()  # an emulated rank fills its own part of the gather only
            # a second drain-and-continue cycle: slots keep alternating
            run.step(STEPS - 1)
            again = run.drain().numpy()
            assert np.array_equal(again[run.mine], rows[run.mine])
            run.close()
        finally:
            eng.close()
    assert (want[:, 6] > 0).all() and np.abs(want[:, 10:].sum(1) - 1).max() < 1e-12
    assert np.array_equal(got, want)


def test_two_real_ranks_share_the_gpu_over_gloo():
    """The NON-emulated multi-rank path of BASELINE configs[3] on real kernels: two ranks (one process each, launched by bench.py itself) shard the 64
    images, both drive the one GPU of the box, the one-step-late exchange is a real collective (gloo through host memory -- RCCL needs one GPU per
    rank).  Rank 0 checks the gathered rows of the last step (every image present, weights sum to 1, finite poses and losses) before it prints its line.
    A launcher / exchange check, not a scaling number."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSAC_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "config3", "--steps", "4", "--warmup", "2", "--hyps", "128",
                          "--height", "96", "--width", "128", "--no-cpu-baseline", "--prewarm-ms", "20"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "sharded round-robin over 2 rank" in line["config"]["workload"]


def test_eight_ranks_over_gloo_refuse_the_strong_number():
    """the driver's own command at 8 ranks -- `bench.py --gpus 8 --steps K --warmup W` -- launched with 8 real processes.  On this one-GPU box
    the ranks can only join over gloo (DSAC_BENCH_BACKEND=gloo; all eight drive the same GPU): the weak line must still be printed with n_gpus = 8, and the
    `strong` object must say `refused` -- never a scaling number measured over something that is not RCCL on eight GPUs.  (Small maps: eight engines share
    one GPU and the host's cores.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSAC_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--hyps", "128", "--height", "96", "--width", "128",
                          "--frames-per-step", "4", "--no-cpu-baseline", "--no-host-driver", "--no-single-frame", "--no-k2-forms", "--prewarm-ms", "10"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert "refused" in line["strong"] and "RCCL" in line["strong"]["refused"]
    assert "speedup" not in line["strong"] and "per_rank_ms" not in line["strong"]

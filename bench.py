#!/usr/bin/env python3
"""bench.py -- hypotheses scored / s over a 640x480 scene-coordinate map (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic frame of BASELINE.json configs[1]
("chess"-like single frame, 256 hypotheses, 640x480 coordinate map, one MI355X):
    K1 sample 256 minimal sets + P3P   ->  K2 reproject all 307 200 points under all 256 poses
    (error images, the score-CNN input of the reference, + fused soft-inlier sums)  ->  K3 softmax.
The frame is resident in HBM before the timed region; every output stays in HBM.  With --gpus N every rank
owns its own frame (images shard across GPUs, no data-path collective: "scaling": "weak").

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel K2 (k_reproject): algorithmic bytes
per launch (12*P + 48*N + 4*N*P + 4*N) over the average launch duration measured with HIP
events on the engine's stream inside the timed region.  `cpu_baseline` is the CPU oracle (a port of the
reference path, g++ -Ofast -fopenmp) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes_k2(N, P, explicit_uv, write_err=True):
    # B_fwd(N,P) = 12 P (xyz f32) + 8 P [explicit uv] + 48 N (R|t f32) + 4 N P (err f32 out) + 4 N (score out)
    return 12 * P + (8 * P if explicit_uv else 0) + 48 * N + (4 * N * P if write_err else 0) + 4 * N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--hyps", type=int, default=256)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSAC_BENCH_STREAMS", "1")),
                    help="engine contexts (HIP streams) per GPU")
    ap.add_argument("--overlap", choices=("pipeline", "gated", "stages", "frames"), default="gated",
                    help="With 2 contexts: 'gated' (default) = frames alternate between two contexts whose K2 launches are serialised by events (dsac_set_k2_events): K1/K3 of "
                         "one frame run under K2 of the other and K2 runs alone.  'frames' = the same without the serialisation: two K2 launches may share the GPU -- 5-10 %% "
                         "more throughput on some boxes, but each K2 launch then takes ~25 %% longer, which lowers roofline.frac (profiles/r01_streams_modes.txt).  'pipeline' = "
                         "one context, dsac_sample_ahead / dsac_score_sampled: K1 of frame i+1 on the context's auxiliary stream under K2/K3 of frame i.  'stages' = stream A "
                         "samples frame i+1 while stream B scores frame i")
    ap.add_argument("--frames-per-step", type=int, default=int(os.environ.get("DSAC_BENCH_FRAMES", "8")),
                    help="independent 640x480 frames (each with --hyps hypotheses) batched into one step: dsac_set_frames / dsac_score_hypotheses_frames carry "
                         "them through K1, K2, K3 in three launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--separate-calls", action="store_true", help="use dsac_sample / dsac_reproject / dsac_softmax instead of the fused call")
    ap.add_argument("--event-stride", type=int, default=8, help="time every n-th K2 launch with HIP events (0 = none)")
    ap.add_argument("--kernel-only", action="store_true", help="time K2 alone on random poses (BASELINE.json configs[2] style)")
    ap.add_argument("--k2-mode", choices=("both", "err", "soft"), default="both", help="K2 outputs: error images and/or soft-inlier sums")
    args = ap.parse_args()

    import torch
    import dsac_amd
    from dsac_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    distributed = world > 1
    backend = os.environ.get("DSAC_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only to exercise the N>1 path on one GPU
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if distributed and backend == "nccl" and local_rank >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the tiny timing tensors are reduced

    N, H, W = args.hyps, args.height, args.width
    P = H * W
    K, Wm = args.steps, args.warmup

    # one frame per rank (seed 1305 + rank: the reference's ThreadRand seed), uploaded before timing; with --frames-per-step B a batch of
    # B different frames per rank
    B = max(1, args.frames_per_step)
    batched = B > 1
    if batched and (args.kernel_only or args.separate_calls or args.k2_mode != "both" or args.overlap == "stages" or N % 128 != 0):
        B, batched = 1, False  # those modes time single frames (K2-only runs, the in-context pipeline, odd hypothesis counts)
    fr = synth.chess_like_frame(H, W, seed=1305 + rank)
    if batched:
        frs = [fr] + [synth.chess_like_frame(H, W, seed=1305 + world * 1000 + rank * B + f) for f in range(1, B)]
        xyz = torch.from_numpy(np.ascontiguousarray(np.stack([f_["xyz"] for f_ in frs]))).to(dev)
    else:
        xyz = torch.from_numpy(fr["xyz"]).to(dev)
    pipelined = (args.overlap == "pipeline" and not args.kernel_only and args.k2_mode == "both" and not args.separate_calls)
    n_ctx = 1 if pipelined else max(1, args.streams)
    n_buf = 2 if pipelined else n_ctx
    engines, bufs = [], []
    for i in range(n_ctx):
        st = torch.cuda.Stream(device=dev)
        eng = dsac_amd.Engine(local_rank, stream=st)
        if batched:
            eng.set_frames(xyz, None, H, W, fr["cam"], borrow=True)
        else:
            eng.set_frame(xyz, None, H, W, fr["cam"], borrow=True)  # implicit full-resolution pixel grid
        eng.profile_enable(args.event_stride > 0, stride=max(1, args.event_stride))
        engines.append((eng, st))
    NB = N * B  # hypotheses per step and context
    for i in range(n_buf):
        bufs.append(dict(
            poses=torch.zeros(NB, 6, dtype=torch.float64, device=dev), sets=torch.zeros(NB, 4, dtype=torch.int32, device=dev),
            ok=torch.zeros(NB, dtype=torch.uint8, device=dev), err=torch.empty(NB, P, dtype=torch.float32, device=dev),
            soft=torch.zeros(NB, dtype=torch.float64, device=dev), w=torch.zeros(NB, dtype=torch.float64, device=dev),
            ent=torch.zeros(B, dtype=torch.float64, device=dev), avg=torch.zeros(B, 6, dtype=torch.float64, device=dev)))
    gated = (n_ctx == 2 and args.overlap == "gated")
    if gated:
        # K2 launches of the two contexts run back to back (never overlapping each other); K1 / K3 of one frame overlap K2 of the other
        evs = [torch.cuda.Event(), torch.cuda.Event()]
        for e in evs:
            e.record(torch.cuda.current_stream(dev))  # materialise the underlying hipEvent_t
        torch.cuda.synchronize(dev)
        engines[0][0].set_k2_events(wait_before=evs[1], record_after=evs[0])
        engines[1][0].set_k2_events(wait_before=evs[0], record_after=evs[1])
    if args.kernel_only:
        rp = synth.random_poses(N, seed=7) + np.array([0, 0, 0, 0, 0, 2500.0])
        for b in bufs:
            b["poses"].copy_(torch.from_numpy(rp))
    torch.cuda.synchronize(dev)

    staged = (n_ctx == 2 and args.overlap == "stages" and not args.kernel_only)
    ev_sampled = [torch.cuda.Event() for _ in range(2)]
    ev_scored = [None, None]

    def step_staged(i):
        # two-stage software pipeline over double-buffered pose sets: K1 of frame i+1 runs under K2/K3 of frame i
        (engA, stA), (engB, stB) = engines
        k = i & 1
        b = bufs[k]
        if ev_scored[k] is not None:
            stA.wait_event(ev_scored[k])       # the scoring stage has finished reading this buffer
        engA.sample(N, seed=1305 + 7919 * i + rank, thr=10.0, max_tries=1 << 16, out=(b["poses"], b["sets"], b["ok"]))
        ev_sampled[k].record(stA)
        stB.wait_event(ev_sampled[k])
        engB.reproject(b["poses"], N=N, clamp=100.0, err=b["err"], soft=b["soft"], tau=10.0, beta=0.5)
        engB.softMax(b["soft"], 0.1, b["poses"], N=N, out=(b["w"], b["ent"], b["avg"]))
        ev_scored[k] = torch.cuda.Event()
        ev_scored[k].record(stB)

    def seed_of(i):
        return 1305 + 7919 * i + rank

    def step_pipelined(i):
        # software pipeline inside ONE context: K1 of frame i+1 (aux stream) under K2/K3 of frame i (main stream)
        eng, _ = engines[0]
        k = i & 1
        nb = bufs[1 - k]
        eng.sampleAhead(1 - k, NB, seed_of(i + 1), nb["poses"], nb["sets"], nb["ok"], thr=10.0, max_tries=1 << 16)
        b = bufs[k]
        eng.scoreSampled(k, b["poses"], b["soft"], b["w"], ent=b["ent"], avg=b["avg"], err=err_shared, clamp=100.0, tau=10.0, beta=0.5, scale=0.1)

    def step(i):
        if pipelined:
            return step_pipelined(i)
        if staged:
            return step_staged(i)
        eng, _ = engines[i % n_ctx]
        b = bufs[i % n_ctx]
        if batched:
            # three launches for B frames: K1 over B*N waves, K2 over B*N error images, K3 with one workgroup per frame
            eng.scoreHypothesesFrames(N, seed=1305 + 7919 * i + rank, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1,
                                      err=b["err"], out=(b["poses"], b["sets"], b["ok"], b["soft"], b["w"], b["ent"], b["avg"]))
            return
        if not args.kernel_only and args.k2_mode == "both" and not args.separate_calls:
            # one C-ABI call: K1 (+ staged pose records) -> K2 -> soft reduce -> K3
            eng.scoreHypotheses(N, seed=1305 + 7919 * i + rank, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1,
                                err=b["err"], out=(b["poses"], b["sets"], b["ok"], b["soft"], b["w"], b["ent"], b["avg"]))
            return
        if not args.kernel_only:
            eng.sample(N, seed=1305 + 7919 * i + rank, thr=10.0, max_tries=1 << 16, out=(b["poses"], b["sets"], b["ok"]))
        eng.reproject(b["poses"], N=N, clamp=100.0, err=b["err"] if args.k2_mode != "soft" else None,
                      soft=b["soft"] if args.k2_mode != "err" else None, tau=10.0, beta=0.5)
        if not args.kernel_only:
            eng.softMax(b["soft"], 0.1, b["poses"], N=N, out=(b["w"], b["ent"], b["avg"]))

    def sync_all():
        for eng, _ in engines:
            eng.synchronize()
        torch.cuda.synchronize(dev)

    if pipelined:
        err_shared = bufs[0]["err"]  # the scoring stage is serial: one error-image buffer
        engines[0][0].sampleAhead(0, NB, seed_of(0), bufs[0]["poses"], bufs[0]["sets"], bufs[0]["ok"], thr=10.0, max_tries=1 << 16)
    for i in range(Wm):
        step(i)
    sync_all()
    for eng, _ in engines:
        eng.profile_read(0, reset=True)

    if distributed:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    for i in range(K):
        step(Wm + i)
    sync_all()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    k2_ms, k2_n = 0.0, 0
    for eng, _ in engines:
        ms, n = eng.profile_read(0, reset=True)
        k2_ms += ms
        k2_n += n
    ok_frac = float(bufs[0]["ok"].float().mean().item()) if not args.kernel_only else 1.0
    wsum = float(bufs[0]["w"][:N].sum().item()) if not args.kernel_only else 1.0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kk = torch.tensor([k2_ms, float(k2_n)], dtype=torch.float64, device=cdev)
        dist.all_reduce(kk, op=dist.ReduceOp.SUM)
        k2_ms, k2_n = float(kk[0].item()), int(kk[1].item())

    if rank == 0:
        total_hyps = N * B * K * world
        value = total_hyps / elapsed
        k2_avg_s = (k2_ms / max(1, k2_n)) * 1e-3
        abytes = B * algorithmic_bytes_k2(N, P, explicit_uv=False, write_err=args.k2_mode != "soft")  # one launch carries B frames
        achieved = abytes / k2_avg_s / 1e9 if k2_avg_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("N") == N * B and tj.get("P") == P and tj.get("frames", 1) == B:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "hypotheses scored/sec over 640x480 coord map",
            "value": value, "unit": "hyp/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: 'chess'-like frame, %d hypotheses, %dx%d coord map%s, %s"
                                    % (N, W, H, (" x %d independent frames per step (one launch each for K1, K2, K3)" % B) if batched else " (single frame per step)", "K2 only on random poses" if args.kernel_only else
                                       "K1 sample+P3P -> K2 reproject (error images + soft-inlier) -> K3 softmax")),
                       "hypotheses_per_frame": N, "frame": [H, W], "frames_per_step": B, "streams_per_gpu": n_ctx,
                       "overlap": ("in-context software pipeline: K1(i+1) || K2,K3(i)" if pipelined else "K2 launches serialised across 2 contexts, K1/K3 overlap them" if gated else "sampling stage || scoring stage" if staged
                                   else ("frames round-robin" if n_ctx > 1 else "none")),
                       "parallelism": "images sharded over %d GPU(s), no data-path collective" % world,
                       "accepted_fraction": ok_frac, "softmax_sum": wsum},
            "roofline": {"kernel": "k_reproject (K2)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": abytes,
                         "avg_launch_us": k2_avg_s * 1e6, "launches_timed": k2_n},
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc
            orc.build()
            cores = orc.num_threads()
            sec1, _ = orc.time_forward(N, 1305, fr["xyz"], fr["uv"], H, W, fr["cam"], reps=1)
            reps = int(max(1, min(64, round(args.cpu_seconds / max(sec1, 1e-3)))))
            sec, _ = orc.time_forward(N, 1305, fr["xyz"], fr["uv"], H, W, fr["cam"], reps=reps)
            out["cpu_baseline"] = {"value": N * reps / sec, "unit": "hyp/s", "cores": cores, "kind": "port",
                                   "sample": "%d frame(s) x %d hypotheses x %dx%d, same workload (sample+P3P, error images, soft-inlier, softmax), "
                                             "oracle built g++ -Ofast -fopenmp, %.1f s" % (reps, N, W, H, sec)}
        print(json.dumps(out), flush=True)

    for eng, _ in engines:
        eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

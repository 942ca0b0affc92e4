#!/usr/bin/env python3
"""bench.py -- hypotheses scored / s over a 640x480 scene-coordinate map (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input of BASELINE.json configs[1]
("chess"-like frame, 256 hypotheses, 640x480 coordinate map, one MI355X), by default 16 independent frames per step:
    K1 sample 256 minimal sets + P3P   ->  K2 reproject all 307 200 points under all 256 poses
    (error images, the score-CNN input of the reference, + fused soft-inlier sums)  ->  K3 softmax.
The frames are resident in HBM before the timed region; every output stays in HBM.

--gpus N: one process per GPU.  Under torchrun (RANK / WORLD_SIZE in the environment) this process is one rank; without it
`python bench.py --gpus N` LAUNCHES the N ranks itself (rank r on GPU r, RCCL) and relays rank 0's line.  Every rank owns its own
frames (images shard across GPUs, no data-path collective: "scaling": "weak"); `n_gpus` is the number of ranks that actually joined.

--workload config3 runs BASELINE.json configs[3] instead: 64 images x 256 hypotheses sharded round-robin over the ranks
(dsac_amd.dist.shard_images), the 64 x (6 + N) results gathered on rank 0 (gather_frame_results) inside the timed region; the
total work is fixed ("scaling": "strong").

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel K2 (k_reproject): algorithmic bytes per launch
(12*P + 48*N + 4*N*P + 4*N per frame) over the average launch duration measured with HIP events on the engine's
stream inside the timed region (every launch when steps <= 64).  `single_frame` repeats the measurement with ONE frame per step (the
literal configs[1]); `rates` separates the kernel-only (K2) rate from the per-image (K1+K2+K3) rate.  `cpu_baseline` is the CPU oracle
(a port of the reference path, g++ -Ofast -fopenmp) timed on this box's host cores on a bounded sample, plus a 1-thread number.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3  # fp32 vector peak: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz (vendor figure)
FLOP_PER_PAIR = 36        # SURVEY.md 8(d), fused soft-inlier mode: 9 FMA + rcp + 2 mul + 2 FMA + 2 sub + 3 + min + ~5 (sigmoid) per (hypothesis, pixel)
CONFIG3_IMAGES = 64    # BASELINE.json configs[3]
# K2's arithmetic forms (include/dsac_hip.h "k2_flags") and what each is tested at (tests/test_gpu_k2_precise.py, tests/test_gpu_k2_exact.py).
# The reference projects in double and rounds the image-plane difference to float once (core/cnn_softam.h:319-362).
K2_FLAG_PRECISE, K2_FLAG_EXACT = 1 << 25, 1 << 28
# form name -> (k2_flags, k2_exact_auto).  Since round 6 the library's auto policy takes the exact-transform form; "fast" switches that off
K2_FORMS = {"exact": (0, 1), "fast": (0, 0), "precise": (K2_FLAG_PRECISE, 1)}
K2_TOLERANCES = {
    "stated": {"residual_px": 1e-3, "softmax_weight": 1e-4, "source": "BASELINE.md 3 / SURVEY.md 8(c)"},
    "fast": {"what": "fp32 pose records, exact-fp32 matrix-core transform (v_mfma_f32_16x16x4_f32), fp32 tail",
             "all_cells_max_px": 4.2e-3, "cells_above_1e-3_px": "3 of 78.6 M (within ~100 mm of a camera centre)", "near_tie_weight_error": 4.4e-3},
    "exact": {"what": "pose records and coordinates as fp16 fixed-point pieces, two fp16 matrix-core accumulations per row (one exact), the camera-frame point "
                      "rounded to float once, the distance as n rsq(n z^2) with n = (pu z - x)^2 + (pv z - y)^2 (one transcendental, none biased per hypothesis), fp32 sigmoid and sums",
              "all_cells_max_px": 5.7e-4, "cells_above_1e-3_px": "0", "near_tie_weight_error": 8.3e-5},
    "precise": {"what": "fp64 records, fp64 transform and perspective division on the vector ALU, one rounding per image-plane difference",
                "all_cells_max_px": 4.6e-5, "cells_above_1e-3_px": "0", "near_tie_weight_error": 4.5e-5},
    "note": "near_tie_weight_error = 0.25 x 0.1 x max |d_i - d_j| of the soft-inlier scores over pairs of UNRELATED hypotheses within 5 % of the top score "
            "(256 hypotheses x 640x480, ~3 900 pairs); figures are the suite's measured worst cases, asserted there with margin",
}


def algorithmic_bytes_k2(N, P, explicit_uv, write_err=True):
    # B_fwd(N,P) = 12 P (xyz f32) + 8 P [explicit uv] + 48 N (R|t f32) + 4 N P (err f32 out) + 4 N (score out)
    return 12 * P + (8 * P if explicit_uv else 0) + 48 * N + (4 * N * P if write_err else 0) + 4 * N


# issue rates of the VALU classes on this chip, wall cycles per wave64 instruction and SIMD at a nominal 2.4 GHz (profiles/r03_valu_rate.txt,
# r03_valu_rate_trans.txt, r03_valu_rate_mfma.txt: measured with scripts/micro/valu_rate.hip; MFMAs do not overlap the VALU stream)
ISSUE_CYCLES = {"packed_fp32": 5.3, "transcendental": 8.8, "plain_valu": 2.9, "mfma": 38.0}
# the exact-transform form issues fp16 matrix-core instructions (K = 32 and K = 16): 17.7 cycles each (profiles/r06_mfma_f64_probe.txt: six alone 102-110 cycles)
ISSUE_CYCLES_EXACT = dict(ISSUE_CYCLES, mfma=17.7)
SOFT_ONLY_FORM = ["exact"]  # the K2 form of the run (set by main): which kernel a soft-inlier-only launch takes


def soft_only_isa_mix(n_hyps, P):
    """Instruction counts per 16 hypotheses x 64 pixels of the kernel the auto policy runs for a soft-inlier-only launch of this size, as
    dsac_amd/csrc/Makefile read them from the built ISA (scripts/isa_mix.py).  None when the launch takes a form that is not priced."""
    nbytes = float(n_hyps) * float(P) * 4.0
    if SOFT_ONLY_FORM[0] == "exact":
        form = 84  # the exact-transform form is one kernel at every size: <64 hypotheses, 256 pixels>, one-wave workgroups
    else:
        form = 58 if nbytes > 4.4e9 else 45 if nbytes > 1.0e9 else None  # k_forward.hip reproject(): the fp32 forms' auto policy by size
    if form is None:
        return None
    path = os.path.join(ROOT, "dsac_amd", "csrc", "build", "k2_soft_isa_%d.json" % form)
    try:
        j = json.load(open(path))
        return dict(j["per_1024_pairs"], kernel=j["kernel"], form=form, source=j["source"])
    except Exception:  # noqa: BLE001 -- not built here: no price rather than a stale constant
        return None


def soft_only_roofline(n_hyps, P, k2_s, launches):
    """SURVEY.md 8(d) secondary measurement: the fused soft-inlier mode (scores without the error-image output) moves 12 P + 48 N + 4 N bytes
    for N P x 36 flop -- it is priced against the fp32 VECTOR roof, never as an HBM fraction."""
    flop = float(n_hyps) * float(P) * FLOP_PER_PAIR
    ach = flop / k2_s / 1e12 if k2_s > 0 else 0.0
    out = {"kernel": "k_reproject (K2), soft-inlier sums only (no error-image output)", "bound": "valu", "achieved": ach, "peak": VALU_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": ach / VALU_PEAK_TFLOPS, "flop_per_launch": flop, "flop_per_pair": FLOP_PER_PAIR, "avg_launch_us": k2_s * 1e6,
           "launches_timed": launches, "hyp_per_s": n_hyps / k2_s if k2_s > 0 else None}
    # issue_model: the kernel's OWN instruction mix per 16 hypotheses x 64 pixels -- counted in the built ISA by the Makefile, so it follows the kernel
    # when the kernel changes -- priced with the issue rates measured on this chip: the time the arithmetic cannot go below
    mix = soft_only_isa_mix(n_hyps, P)
    if mix is not None:
        rates = ISSUE_CYCLES_EXACT if mix["form"] == 84 else ISSUE_CYCLES
        cyc = sum(mix[k] * rates[k] for k in rates)
        priced_s = float(n_hyps) * float(P) / 1024.0 * cyc / (1024 * 2.4e9)
        out["issue_model"] = {"instructions_per_1024_pairs": {k: mix[k] for k in rates}, "cycles_per_instruction": rates,
                              "cycles_per_1024_pairs": cyc, "priced_us": priced_s * 1e6, "frac": priced_s / k2_s if k2_s > 0 else None,
                              "kernel_form": mix["form"], "source": mix["source"] + "; issue rates: profiles/r03_valu_rate*.txt"}
    return out


def event_stride_for(steps, requested):
    """Every K2 launch is timed when the run is short (the driver's --steps 20 would otherwise leave 3 samples); long runs time
    about 64 launches spread over the region (event records are not free on the stream)."""
    if requested is not None and requested >= 0:
        return requested
    return 1 if steps <= 64 else max(1, steps // 64)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--hyps", type=int, default=256)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--workload", choices=("frames", "config3", "config5"), default="frames",
                    help="frames: BASELINE.json configs[1] frames, --frames-per-step per step and rank (weak scaling).  config3: configs[3], 64 images x "
                         "--hyps hypotheses sharded over the ranks, every image through the whole processImage (sample, score, soft-argmax, 8 refinement "
                         "steps, loss), refined poses + losses + weights gathered (strong scaling); a step is one pass over the 64 images.  config5: "
                         "configs[4], one end-to-end training step per rank and step (one frame per GPU, both CNNs, geometry forward + backward, the "
                         "gradient exchange of ~157 MB over RCCL), weak scaling")
    ap.add_argument("--reduce-mode", choices=("all_reduce", "reduce_scatter"), default="all_reduce",
                    help="config5: how a gradient bucket travels (reduce_scatter = reduce-scatter + all-gather over all xGMI links, RCCL only)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSAC_BENCH_STREAMS", "1")),
                    help="engine contexts (HIP streams) per GPU")
    ap.add_argument("--overlap", choices=("none", "pipeline", "gated", "stages", "frames"), default=os.environ.get("DSAC_BENCH_OVERLAP", "gated"),
                    help="With 2 contexts: 'gated' (default) = frames alternate between two contexts whose K2 launches are serialised by events (dsac_set_k2_events): K1/K3 of "
                         "one frame run under K2 of the other and K2 runs alone.  'frames' = the same without the serialisation.  'pipeline' = one context, "
                         "dsac_sample_ahead / dsac_score_sampled: K1 of step i+1 on the context's auxiliary stream under K2/K3 of step i, consecutive steps on DIFFERENT "
                         "frames.  'stages' = stream A samples frame i+1 while stream B scores frame i.  With one context and not 'pipeline': no overlap")
    ap.add_argument("--frames-per-step", type=int, default=int(os.environ.get("DSAC_BENCH_FRAMES", "16")),
                    help="independent 640x480 frames (each with --hyps hypotheses) batched into one step: dsac_set_frames / dsac_score_hypotheses_frames carry "
                         "them through K1, K2, K3 in three launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-driver", action="store_true", help="skip the timing of the C++ evaluation program (dsac_amd/host/test_ransac_softam)")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the extra single-frame (literal configs[1]) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed clock / TLB settling before the warm-up steps (not counted as steps)")
    ap.add_argument("--separate-calls", action="store_true", help="use dsac_sample / dsac_reproject / dsac_softmax instead of the fused call")
    ap.add_argument("--event-stride", type=int, default=-1, help="time every n-th K2 launch with HIP events (0 = none, -1 = every launch up to 64 steps)")
    ap.add_argument("--kernel-only", action="store_true", help="time K2 alone on random poses (BASELINE.json configs[2] with --hyps 4096)")
    ap.add_argument("--k2-mode", choices=("both", "err", "soft"), default="both", help="K2 outputs: error images and/or soft-inlier sums")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="default workload on ONE GPU: fill the line's `strong` object (configs[3], 64 images fixed) from the emulation of one rank of W.  "
                         "config3 on ONE GPU: besides the whole 64-image step, time exactly the share rank --emulate-rank of W ranks would run (8 images at "
                         "W = 8; its gather replaced by the rank's own part) and report per_rank_ms next to one_gpu_ms / W -> predicted efficiency")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--k2-form", choices=tuple(K2_FORMS), default=os.environ.get("DSAC_BENCH_K2_FORM", "exact"),
                    help="the arithmetic form of K2 the line's `value` is measured on: exact (the library's default since round 6: every stated tolerance holds), "
                         "fast (fp32 matrix-core transform, rounds 2-5), precise (fp64 on the vector ALU)")
    ap.add_argument("--no-k2-forms", action="store_true", help="skip the measurement of K2's other arithmetic forms (`k2_forms`)")
    ap.add_argument("--dry-run", action="store_true", help="exercise launch / shard / gather / reporting without touching a GPU (CPU tests, gloo)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment
# ---------------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Spawn n ranks of this script (rank r -> LOCAL_RANK r -> GPU r), wait for all of them, propagate failure.  Rank 0 prints the JSON
    line on the inherited stdout.  Returns the exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DSAC_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC, needed by RCCL on these hosts
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("DSAC_BENCH_LAUNCH_TIMEOUT", "1800"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is not None:
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in pending:  # one rank failed: the others would wait for it in a collective until the timeout
                        q.terminate()
        if time.time() > deadline:
            for q in pending:
                q.kill()
            return 124
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------------------------------------
BACKEND_NOTE = None  # set when the timing collectives had to fall back from RCCL to gloo


def init_distributed(args):
    """Returns (rank, local_rank, world, backend, dist or None).  world is what actually joined the process group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): refusing to report a rank count that is not the one running" % (world, args.gpus))
    backend = os.environ.get("DSAC_BENCH_BACKEND", "gloo" if args.dry_run else "nccl")  # "nccl" is RCCL on ROCm
    if world == 1:
        return rank, local_rank, 1, backend, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        ndev = torch.cuda.device_count()
        if local_rank >= ndev:
            raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible (one rank per GPU)" % (local_rank, ndev))
        torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            probe = torch.ones(1, device="cuda:%d" % local_rank)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("RCCL probe all-reduce returned %r for %d ranks" % (probe.item(), world))
        except Exception as e:  # noqa: BLE001 -- RCCL unusable on this node
            if os.environ.get("DSAC_BENCH_NO_FALLBACK") or args.workload in ("config3", "config5"):
                raise  # configs[3] gathers its results and configs[4] exchanges its gradients with the backend: no silent change there
            # The default workload has NO data-path collective (images shard, every rank scores its own): RCCL only carries the barrier
            # and the max of the timing scalars.  Rather than lose the scaling line, carry those over gloo and say so in the JSON.
            global BACKEND_NOTE
            BACKEND_NOTE = "timing barrier / max over gloo: RCCL init failed on this node (%s: %s)" % (type(e).__name__, str(e).splitlines()[0][:160])
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            backend = "gloo"
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    joined = dist.get_world_size()
    if joined != args.gpus:
        raise SystemExit("%d ranks joined, --gpus %d" % (joined, args.gpus))
    return rank, local_rank, joined, backend, dist


def cpu_baseline(args, fr, N, H, W):
    """The oracle (port of the reference path) on the host cores: as many threads as the box really grants (effective_cpus), and one thread."""
    from oracle import oracle as orc
    orc.build()
    cores, cpu_desc = orc.effective_cpus()
    orc.set_num_threads(cores)
    sec1, _ = orc.time_forward(N, 1305, fr["xyz"], fr["uv"], H, W, fr["cam"], reps=1)
    reps = int(max(1, min(64, round(args.cpu_seconds / max(sec1, 1e-3)))))
    sec, _ = orc.time_forward(N, 1305, fr["xyz"], fr["uv"], H, W, fr["cam"], reps=reps)
    out = {"value": N * reps / sec, "unit": "hyp/s", "cores": cores, "cpus": cpu_desc, "kind": "port",
           "sample": "%d frame(s) x %d hypotheses x %dx%d, same workload (sample+P3P, error images, soft-inlier, softmax), "
                     "oracle built g++ -Ofast -fopenmp, %.1f s" % (reps, N, W, H, sec)}
    # one thread (deterministic RNG stream order); bounded: a quarter of the hypotheses of one frame
    n1 = max(16, N // 4)
    orc.set_num_threads(1)
    try:
        s1, _ = orc.time_forward(n1, 1305, fr["xyz"], fr["uv"], H, W, fr["cam"], reps=1)
    finally:
        orc.set_num_threads(cores)
    out["one_thread"] = {"value": n1 / s1, "unit": "hyp/s", "cores": 1, "sample": "%d hypotheses x %dx%d, %.1f s" % (n1, W, H, s1)}
    # the reference's own map size (40 x 40, int16-quantised coordinates), all threads
    from dsac_amd import synth as _synth
    f40 = _synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
    r40 = 20
    s40, _ = orc.time_forward(N, 1305, f40["xyz"], f40["uv"], 40, 40, f40["cam"], reps=r40)
    out["reference_size"] = {"value": N * r40 / s40, "unit": "hyp/s", "cores": cores, "sample": "%d frames x %d hypotheses x 40x40 int16 map, %.2f s" % (r40, N, s40)}
    # the REAL reference code (core/cnn_softam.h processImage, compiled from /root/reference by oracle/refbuild against OpenCV stand-ins; the prebuilt
    # oracle/_ref/*.so travels with the repository) on the same frame size: whole processImage per image -- sampling, 256 error images, soft-inlier score,
    # softmax, 8 refinement steps, loss.  Context only: the stand-ins are not OpenCV's speed.
    try:
        from oracle import reference as _ref
        if _ref.available():
            _ref.lib()
            _ref.set_score_model(10.0, 0.5, 0.1)
            gt40 = _synth.cv_to_jp6(f40["gt_pose"])
            _ref.processImage(1305, f40["xyz"], gt40, hyps=N)
            nref = 4
            t0 = time.perf_counter()
            for i in range(nref):
                _ref.processImage(1306 + i, f40["xyz"], gt40, hyps=N)
            sref = (time.perf_counter() - t0) / nref
            out["reference_build"] = {"kind": "reference", "ms_per_image": sref * 1e3, "value": N / sref, "unit": "hyp/s",
                                      "sample": "%d images x %d hypotheses x 40x40, the reference's own processImage (oracle/_ref/libdsac_ref.so: core/cnn_softam.h "
                                                "against OpenCV stand-ins), as many threads as its own OpenMP pragmas take" % (nref, N)}
    except Exception as e:  # noqa: BLE001  (the reference build is optional context)
        out["reference_build"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def host_driver_leg(N, H, W, device=0, images=64, batch=16, passes=6):
    """The reference-shaped C++ program on the engine (dsac_amd/host/test_ransac_softam: GlobalProperties, FrameBatch over the C ABI, the reference's
    output files) on `images` synthetic frames: one engine context for the run, the data set resident in HBM, `batch` images per launch chain, the
    refinement tail of a batch under the next batch.  Returns what its "Timing:" line says -- the C++ host path's own clock around a pass."""
    import re
    import tempfile
    exe = os.path.join(ROOT, "dsac_amd", "host", "test_ransac_softam")
    if not os.path.exists(exe):
        return {"error": "dsac_amd/host/test_ransac_softam is not built"}
    with tempfile.TemporaryDirectory() as tmp:
        # -warmup 300: untimed passes until the clock has settled (the program makes its synthetic data set on the host first: the GPU idles for a second)
        cmd = [exe, "-synth", str(images), "-mw", str(W), "-mh", str(H), "-rI", str(N), "-batch", str(batch), "-passes", str(passes), "-warmup", "300", "-dev", str(device)]
        try:
            out = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600)
        except Exception as e:  # noqa: BLE001
            return {"error": "%s: %s" % (type(e).__name__, e)}
        m = re.search(r"Timing: .*?: ([0-9.eE+-]+) us per image \(([0-9.eE+-]+) ms per pass over (\d+) timed pass\(es\); first pass ([0-9.eE+-]+) ms; upload \+ set-up ([0-9.eE+-]+) ms\)",
                      out.stdout)
        acc = re.search(r"accuracy: ([0-9.eE+-]+)%", out.stdout)
        if out.returncode != 0 or not m:
            return {"error": "rc %d: %s" % (out.returncode, (out.stdout + out.stderr)[-300:])}
        res = {"program": "dsac_amd/host/test_ransac_softam " + " ".join(cmd[1:]),
               "what": "C++ host over the C ABI (dsac::Context + FrameBatch::processImages = dsac_set_frames + dsac_process_images): the whole processImage of "
                       "every image (sample + P3P, error images, soft-argmax, 8 refinement steps, loss), %d images per launch chain, score and refinement "
                       "tail of a chain under the next one, data set resident in HBM, timed passes enqueued back to back, results copied back once" % batch,
               "us_per_image": float(m.group(1)), "ms_per_pass": float(m.group(2)), "timed_passes": int(m.group(3)), "first_pass_ms": float(m.group(4)),
               "upload_and_setup_ms": float(m.group(5)), "images": images, "hypotheses": N, "accuracy_percent": float(acc.group(1)) if acc else None}
        # the training program in its device-resident form: the training set in HBM, `batch` random frames per round gathered device-to-device, forward
        # (processImage with the error images) and backward (train_ransac_softam.cpp:288-394) as one launch chain each, gradients left in HBM
        exe_t = os.path.join(ROOT, "dsac_amd", "host", "train_ransac_softam")
        if os.path.exists(exe_t):
            cmd_t = [exe_t, "-synth", "32", "-mw", str(W), "-mh", str(H), "-rI", str(N), "-rounds", "60", "-batch", str(batch), "-gradstats", "0", "-warmup", "300",
                     "-dev", str(device)]
            try:
                out_t = subprocess.run(cmd_t, cwd=tmp, capture_output=True, text=True, timeout=600)
                mt = re.search(r"Timing: (\d+) rounds x (\d+) frames .*?: ([0-9.eE+-]+) us per round = ([0-9.eE+-]+) us per frame", out_t.stdout)
                res["training"] = ({"program": "dsac_amd/host/train_ransac_softam " + " ".join(cmd_t[1:]), "rounds": int(mt.group(1)), "frames_per_round": int(mt.group(2)),
                                    "us_per_round": float(mt.group(3)), "us_per_frame": float(mt.group(4)),
                                    "what": "forward + backward of every frame, device-resident (FrameBatch::gatherFramesFrom / processImages / backward)"}
                                   if (out_t.returncode == 0 and mt) else {"error": "rc %d: %s" % (out_t.returncode, (out_t.stdout + out_t.stderr)[-300:])})
                # the same rounds without the error images (-errimg 0): the built-in soft-inlier score does not need them (a score CNN does), and they are what
                # separates this program's round from the Python geometry bench (scripts/train_geometry_bench.py runs K2 sums-only): rocprofv3 of both,
                # profiles/r05_train_gap.txt -- K2 921 against 717 us per 16 frames, the gather of the round's frames 27 us, the rest K4 on other frames
                out_n = subprocess.run(cmd_t + ["-errimg", "0"], cwd=tmp, capture_output=True, text=True, timeout=600)
                mn = re.search(r"Timing: (\d+) rounds x (\d+) frames .*?: ([0-9.eE+-]+) us per round = ([0-9.eE+-]+) us per frame", out_n.stdout)
                if out_n.returncode == 0 and mn and "training" in res and "us_per_frame" in res["training"]:
                    res["training"]["us_per_frame_without_error_images"] = float(mn.group(4))
            except Exception as e:  # noqa: BLE001
                res["training"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return res


def strong_scaling_leg(args, rank, local_rank, world, backend, dist, emulate_world=0, emulate_rank=0, err_buffer=None, engine=None, stream=None):
    """north_star's multi-GPU claim inside the command the driver runs: BASELINE.json configs[3] -- 64 images x --hyps hypotheses FIXED, sharded round-robin
    over the ranks (core/test_ransac_softam.cpp:97-230: independent images), every image through the whole processImage, the 64 x (10 + N) result rows
    exchanged with ONE all_gather_into_tensor per step over RCCL inside the timed region (dsac_amd.shard.ShardRunner).  Returns the `strong` object of
    the JSON line (rank 0; None elsewhere):
        one_gpu_ms   rank 0 ALONE runs all 64 images per step (same process, same engine, before the ranks run their shares; the others wait)
        per_rank_ms  every rank runs its share, barrier + max over ranks, the gather of step i beside step i + 1 as in production
        speedup = one_gpu_ms / per_rank_ms, efficiency = speedup / ranks
        collective_bytes_per_step, collective_exposed_us (the same steps with the gather replaced by the copy of the rank's own part, subtracted)
    world == 1 with emulate_world > 1: the same fields from the one-GPU emulation of rank `emulate_rank` of `emulate_world` (a prediction, flagged)."""
    import torch
    import dsac_amd
    from dsac_amd import synth
    from dsac_amd.shard import ShardRunner
    N, H, W = args.hyps, args.height, args.width
    P = H * W
    if N % 128 != 0:
        return {"refused": "configs[3] needs --hyps to be a multiple of 128"} if rank == 0 else None
    allow_any = bool(os.environ.get("DSAC_BENCH_STRONG_ALLOW_GLOO"))  # tests: two ranks sharing ONE GPU over gloo exercise the leg's control flow
    if world > 1 and backend != "nccl" and not allow_any:
        # the claim is about RCCL over xGMI: a run whose ranks talk over gloo measures something else -- no number rather than a wrong one
        return {"refused": "the ranks did not join an RCCL process group (backend %s%s): no strong-scaling measurement" %
                           (backend, ("; " + BACKEND_NOTE) if BACKEND_NOTE else "")} if rank == 0 else None
    dev = torch.device("cuda", local_rank)
    # the engine (and stream) of the weak run when the caller has one: a SECOND engine in the process measured 15-25 % slower steps (4.41 / 0.62 ms against
    # 3.82 / 0.51 for the same runner on the first engine, gpurun_out/r05b) -- its tail stream did not overlap its main stream; not isolated further
    own_engine = engine is None
    st = stream if stream is not None else torch.cuda.Stream(device=dev)
    eng = engine if engine is not None else dsac_amd.Engine(local_rank, stream=st)
    cam = synth.chess_like_frame(8, 8, seed=1)["cam"]
    perm3 = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    cache = {}

    def frames_of(i):
        if i not in cache:
            cache[i] = synth.chess_like_frame(H, W, seed=1305 + i)["xyz"]  # seeds 1305 + i
        return cache[i]
    B = max(1, args.frames_per_step)
    K = max(args.steps, 20)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the small timing / status tensors are reduced

    def rmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(ok, what):
        """Every phase ends with an agreement: a rank that failed must not leave the others waiting in the next collective.  Returns the reason when any
        rank failed (the same on every rank), else None."""
        bad = rmax(0.0 if ok else 1.0)
        return None if bad == 0.0 else "%s failed on at least one rank%s" % (what, (": " + failure[0]) if failure[0] else "")

    failure = [None]

    def guarded(fn):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 -- reported in the line; the weak-scaling value must still be printed
            failure[0] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            return None

    host_enqueue = [0.0]

    def timed(runner, base, together=True):
        """settle (--prewarm-ms), then K steps + drain (between barriers when the ranks run together): seconds per step on this rank, the last step's rows"""
        t_pre, n = time.perf_counter(), 0
        while n < max(5, args.warmup) or (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            runner.step(base + (n % 90))
            n += 1
            if n % 16 == 0:
                eng.synchronize()
        runner.drain()
        if together:
            barrier()
        else:
            eng.synchronize()
        for key in runner.host_us:
            runner.host_us[key] = 0.0
        t0 = time.perf_counter()
        for i in range(K):
            runner.step(base + 100 + i)
        host_enqueue[0] = (time.perf_counter() - t0) / K  # the host never waits inside a step: this is what enqueueing one costs
        rows = runner.drain()
        eng.synchronize()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / K, rows

    out = None
    # ---- one GPU, all 64 images: rank 0 alone (the other ranks' GPUs idle; they wait at the agreement below)
    one_s, one_host, err_shared = None, 0.0, err_buffer
    if rank == 0:
        def solo():
            r1 = ShardRunner(eng, st, dev, frames_of, CONFIG3_IMAGES, 0, 1, N, H, W, cam, perm3, batch=B, emulate=True, defer=2, err_buffer=err_buffer)
            t, _ = timed(r1, 0, together=False)
            e = r1.err
            r1.close()
            return t, e
        got = guarded(solo)
        if got is not None:
            one_s, err_shared = got
            one_host = host_enqueue[0]
    why = all_ok(one_s is not None or rank != 0, "the one-GPU run of all 64 images")
    if why:
        return {"error": why} if rank == 0 else None
    if world > 1:
        # ---- every rank its share, the real exchange.  The runner is built under the guard too (a rank that cannot allocate must say so before the
        # others enter the first collective)
        rr = guarded(lambda: ShardRunner(eng, st, dev, frames_of, CONFIG3_IMAGES, rank, world, N, H, W, cam, perm3, batch=B, defer=2, err_buffer=err_shared))
        why = all_ok(rr is not None, "setting up the sharded run")
        if why:
            return {"error": why} if rank == 0 else None
        barrier()
        per_s, rows = timed(rr, 1000)
        host_ex = host_enqueue[0]
        barrier()
        per_s = rmax(per_s)
        host_ex_max = rmax(host_ex)  # the slowest rank's host: N processes share the node's cores (a 16-core cgroup for 8 ranks), and a step is ~0.5 ms
        per_rank, Dw = rr.ex.per, rr.ex.D
        rr.close()
        # ---- the same steps without the collective (the rank's own part copied instead): what the gather adds to a step
        re_ = ShardRunner(eng, st, dev, frames_of, CONFIG3_IMAGES, rank, world, N, H, W, cam, perm3, batch=B, emulate=True, defer=2, err_buffer=err_shared)
        barrier()
        noex_s, _ = timed(re_, 2000)
        barrier()
        noex_s = rmax(noex_s)
        re_.close()
        if rank == 0:
            ws = rows[:, 10:].sum(1)
            ok = bool(((ws - 1.0).abs() < 1e-9).all()) and bool(torch.isfinite(rows[:, :10]).all()) and bool((rows[:, 6] > 0).all())
            # rank-count independence: the gathered rows of the sharded run are the rows of the one-GPU run of the same step (same seeds per image)
            out = {"workload": "BASELINE.json configs[3]: %d images x %d hypotheses x %dx%d FIXED, image i on rank i mod %d, whole processImage per image, "
                               "result rows (refined pose 6 + loss 4 + N weights) exchanged by one all_gather_into_tensor per step" % (CONFIG3_IMAGES, N, W, H, world),
                   "ranks_joined": world, "backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else "%s -- a TEST of the leg, not a measurement" % backend,
                   "steps": K, "images_per_rank_step": CONFIG3_IMAGES // world,
                   "one_gpu_ms": one_s * 1e3, "per_rank_ms": per_s * 1e3, "speedup": one_s / per_s, "efficiency": one_s / per_s / world,
                   "collective_bytes_per_step": int(world * per_rank * Dw * 8), "per_rank_ms_without_collective": noex_s * 1e3,
                   "collective_exposed_us": max(0.0, (per_s - noex_s) * 1e6), "rows_ok": ok, "host_enqueue_ms_per_step_rank0": host_ex * 1e3,
                   "host_enqueue_ms_per_step": host_ex_max * 1e3, "host_bound": bool(host_ex_max > 0.8 * per_s),
                   "how": "one_gpu_ms: rank 0 alone, all 64 images per step, before the ranks ran their shares (same process and engine); per_rank_ms: every rank "
                          "its share, barrier + max over ranks, %d steps, the gather of step i on a side stream beside step i + 1" % K}
    elif emulate_world > 1:
        Wem, rem = emulate_world, emulate_rank % emulate_world
        em = ShardRunner(eng, st, dev, frames_of, CONFIG3_IMAGES, rem, Wem, N, H, W, cam, perm3, batch=B, emulate=True, defer=2, err_buffer=err_shared)
        per_s, rows = timed(em, 1000)
        out = {"workload": "BASELINE.json configs[3]: %d images x %d hypotheses x %dx%d FIXED; ONE GPU runs exactly the share of rank %d of %d (its images, "
                           "buffers, launch sequence, deferred tails; the all-gather replaced by the copy of its own part)" % (CONFIG3_IMAGES, N, W, H, rem, Wem),
               "emulated": True, "ranks_joined": 1, "backend": "none (one process: a PREDICTION of what %d ranks would show, not a measurement)" % Wem,
               "steps": K, "images_per_rank_step": len(em.mine), "one_gpu_ms": one_s * 1e3, "per_rank_ms": per_s * 1e3, "speedup": one_s / per_s,
               "efficiency": one_s / per_s / Wem, "collective_bytes_per_step": int(Wem * em.ex.per * em.ex.D * 8), "collective_exposed_us": None,
               "rows_ok": bool(torch.isfinite(rows[em.mine]).all()), "host_enqueue_ms_per_step": host_enqueue[0] * 1e3, "one_gpu_host_enqueue_ms_per_step": one_host * 1e3,
               "host_enqueue_us_by_phase": {k_: v_ / K * 1e6 for k_, v_ in em.host_us.items()}}
        em.close()
    elif rank == 0:
        out = {"workload": "BASELINE.json configs[3] on one GPU", "ranks_joined": 1, "one_gpu_ms": one_s * 1e3, "steps": K}
    if own_engine:
        eng.close()
    return out


def dry_strong(args, rank, world, dist):
    """The `strong` object of the line with a stand-in for the engine (CPU, gloo): the exchange, the rank-0-alone run, the max over ranks and the field
    set are the real code paths of strong_scaling_leg; the "work" of an image is a sleep.  Returns the object on rank 0."""
    import torch
    from dsac_amd import dist as ddist
    N, K = args.hyps, max(args.steps, 3)
    per_image_s = 2e-4

    def run(r, w, group):
        mine = ddist.shard_images(CONFIG3_IMAGES, r, w)
        ex = ddist.FrameResultExchange(CONFIG3_IMAGES, r, w, (6, 4, N), torch.device("cpu"), group=group, local_only=group is None)
        t0 = time.perf_counter()
        for i in range(K):
            k = i & 1
            if i >= 1:
                ex.wait(k)
                ex.launch(1 - k)
            ref_v, out4_v, w_v = ex.views(k)
            for j, img in enumerate(mine):
                ref_v[j] = float(img)
                out4_v[j] = float(i)
                w_v[j] = 1.0 / N
            time.sleep(per_image_s * len(mine))
        k = (K - 1) & 1
        if K >= 2:
            ex.wait(1 - k)
        ex.launch(k)
        ex.wait(k)
        return (time.perf_counter() - t0) / K, ex.frames(k), ex
    one_s = None
    if rank == 0:
        one_s, _, _ = run(0, 1, None)
    if dist is not None:
        dist.barrier()
    Wsh = world if world > 1 else args.emulate_world
    per_s, rows, ex = run(rank if world > 1 else args.emulate_rank % Wsh, Wsh, (dist.group.WORLD if world > 1 else None))
    if dist is not None:
        t = torch.tensor([per_s], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_s = float(t.item())
    if rank != 0:
        return None
    ok = bool(torch.equal(rows[:, 0], torch.arange(CONFIG3_IMAGES, dtype=torch.float64))) if world > 1 else True
    host_field = {"host_enqueue_ms_per_step": per_s * 1e3} if world > 1 else {}  # the stand-in engine IS the host: the field of the real leg, max over ranks
    return {"workload": "DRY RUN of BASELINE.json configs[3] (stand-in engine): %d images fixed over %d rank(s)" % (CONFIG3_IMAGES, Wsh), "dry_run": True,
            "emulated": world == 1, "ranks_joined": world, "backend": ("gloo (dry run)" if world > 1 else "none"), "steps": K,
            "images_per_rank_step": CONFIG3_IMAGES // Wsh, "one_gpu_ms": one_s * 1e3, "per_rank_ms": per_s * 1e3, "speedup": one_s / per_s,
            "efficiency": one_s / per_s / Wsh, "collective_bytes_per_step": int(Wsh * ex.per * ex.D * 8), "collective_exposed_us": None, "rows_ok": ok, **host_field}


def run_dry(args, rank, world, dist):
    """No GPU: the rank/shard/gather plumbing and the JSON contract with a stand-in for the engine (CPU tests)."""
    import torch
    from dsac_amd import dist as ddist
    N = args.hyps
    K = args.steps
    if args.workload == "config5":
        return run_config5(args, rank, 0, world, "gloo", dist)
    if args.workload == "config3":
        # the exchange of the real run (dsac_amd.shard.ShardRunner) with a stand-in for the engine: every step writes its rows into the slot's regions,
        # the gather of step i is launched at the top of step i + 1 and consumed at the top of step i + 2
        mine = ddist.shard_images(CONFIG3_IMAGES, rank, world)
        ex = ddist.FrameResultExchange(CONFIG3_IMAGES, rank, world, (6, 4, N), torch.device("cpu"))
        t0 = time.perf_counter()
        for i in range(K):
            k = i & 1
            if i >= 1:
                ex.wait(k)
                ex.launch(1 - k)
            ref_v, out4_v, w_v = ex.views(k)
            for j, img in enumerate(mine):
                ref_v[j] = float(img)
                out4_v[j] = float(i)
                w_v[j] = 1.0 / N
        k = (K - 1) & 1
        if K >= 2:
            ex.wait(1 - k)
        ex.launch(k)
        ex.wait(k)
        allres = ex.frames(k)
        elapsed = time.perf_counter() - t0
        assert torch.equal(allres[:, 0], torch.arange(CONFIG3_IMAGES, dtype=torch.float64)), "gather lost or misplaced frames"
        assert bool((allres[:, 6] == float(K - 1)).all()) and bool(((allres[:, 10:].sum(1) - 1.0).abs() < 1e-12).all()), "gather returned a stale step"
        total = CONFIG3_IMAGES * N * K
        scaling, per_step = "strong", CONFIG3_IMAGES
    else:
        t0 = time.perf_counter()
        time.sleep(0.01)
        elapsed = time.perf_counter() - t0
        total = N * args.frames_per_step * K * world
        scaling, per_step = "weak", args.frames_per_step * world
    strong = None
    if args.workload == "frames" and (world > 1 or args.emulate_world > 1):
        strong = dry_strong(args, rank, world, dist)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "hypotheses scored/sec over 640x480 coord map", "value": total / elapsed, "unit": "hyp/s", "n_gpus": world, "steps": K,
                          "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN (no GPU work): %s" % args.workload, "frames_per_step_all_ranks": per_step},
                          **({"strong": strong} if strong is not None else {})}), flush=True)


def run_config5(args, rank, local_rank, world, backend, dist):
    """BASELINE.json configs[4] / SURVEY.md 8(d) config 5: one end-to-end training step per rank and step -- one frame per GPU
    (core/train_ransac_softam.cpp:227-235 trains on one image per step), scene-coordinate CNN and score CNN with the reference's architectures and
    random weights, the geometry forward + backward on the engine (K1 ... K7, K4), the gradient exchange of both CNNs (~157 MB fp32) launched from
    inside the backward passes (dsac_amd.dist.GradientReducer) and waited for at the optimizer step.  Reports the step, its geometry share, the
    stand-alone cost of the exchange and how much of it the step still shows.  --dry-run: small CPU networks over gloo (launcher / reducer /
    reporting test), no GPU."""
    import torch
    from dsac_amd import dist as ddist
    K, Wm, N = args.steps, args.warmup, args.hyps
    distributed = world > 1

    def barrier(devsync=None):
        if devsync is not None:
            torch.cuda.synchronize(devsync)
        if distributed:
            dist.barrier()

    def rmax(x, cdev):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.dry_run:
        torch.manual_seed(rank)
        nets = [torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 3)),
                torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 1))]
        reds = [ddist.GradientReducer(n.parameters(), bucket_bytes=16 << 10, mode=args.reduce_mode) for n in nets]
        x = [torch.randn(16, 64), torch.randn(16, 32)]
        cdev = torch.device("cpu")
        barrier()
        t0 = time.perf_counter()
        ncoll = 0
        for _ in range(K):
            for n_, x_ in zip(nets, x):
                for p in n_.parameters():
                    p.grad = None if p.grad is None else p.grad.zero_()
                n_(x_).sum().backward()
            ncoll = sum(r.wait() for r in reds)
        barrier()
        step_s = rmax((time.perf_counter() - t0) / K, cdev)
        grads = [p.grad for n_ in nets for p in n_.parameters()]
        t0 = time.perf_counter()
        for _ in range(K):
            ddist.all_reduce_gradients(grads, bucket_bytes=16 << 10)
        coll_s = rmax((time.perf_counter() - t0) / K, cdev)
        nparam = sum(g.numel() for g in grads)
        line = dict(step_ms=step_s * 1e3, geometry_ms=None, cnn_ms=None, collective_ms=coll_s * 1e3 if distributed else 0.0, collectives_per_step=ncoll,
                    grad_bytes=4 * nparam, dry_run=True)
        value = N * world / step_s
        dev_name = "cpu"
    else:
        from dsac_amd import e2e, synth
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        di = local_rank % ndev
        torch.cuda.set_device(di)
        ts = e2e.TrainStep(di, hyps=N, sub_sample=0.01, reduce_mode=args.reduce_mode)
        dev = ts.dev
        cdev = dev if backend == "nccl" else torch.device("cpu")
        fr = synth.chess_like_frame(40, 40, seed=1305 + rank, quantise_int16=True)  # the reference's 40 x 40 stratified sub-sample (core/lua_calls.h:33)
        patches = torch.rand(1600, 3, 42, 42, device=dev) * 255
        uv = torch.as_tensor(fr["uv"], device=dev)
        off = torch.as_tensor(fr["xyz"], device=dev)  # random-weight CNN predicts ~0: the synthetic scene rides on an offset, the gradient path is the real one
        perm = synth.fast_permutations(1600, 8)
        gt = synth.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
        for i in range(Wm):
            ts.step(patches, uv, gt, perm, seed=1000 + i, xyz_offset_mm=off)
        barrier(dev)
        t0 = time.perf_counter()
        for i in range(K):
            out = ts.step(patches, uv, gt, perm, seed=2000 + i, xyz_offset_mm=off)
        barrier(dev)
        step_s = rmax((time.perf_counter() - t0) / K, cdev)
        # the same steps with the segment events on: geometry vs CNN time on the stream (events cost a few us each; not part of step_ms)
        ts.timing = True
        geo = cnn = 0.0
        for i in range(5):
            ts.step(patches, uv, gt, perm, seed=3000 + i, xyz_offset_mm=off)
            torch.cuda.synchronize(dev)
            geo += ts.last_segments_ms["geometry"] / 5
            cnn += ts.last_segments_ms["cnn"] / 5
        ts.timing = False
        # the exchange alone (no overlap): launch + wait of the same buckets on the gradients that are there
        coll_s = 0.0
        if distributed:
            grads = [p.grad for p in ts.params()]
            for _ in range(2):
                ddist.all_reduce_gradients(grads, mode=args.reduce_mode)
            barrier(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                ddist.all_reduce_gradients(grads, mode=args.reduce_mode)
            barrier(dev)
            coll_s = rmax((time.perf_counter() - t0) / 10, cdev)
        # ... and the step without any exchange (what the exchange adds to the step = what is NOT hidden under the backward)
        local_s = step_s
        if distributed:
            ts.reducer_score.enabled = ts.reducer_obj.enabled = False
            for h in ts.reducer_score._hooks + ts.reducer_obj._hooks:
                h.remove()
            barrier(dev)
            t0 = time.perf_counter()
            for i in range(K):
                ts.step(patches, uv, gt, perm, seed=4000 + i, xyz_offset_mm=off)
            barrier(dev)
            local_s = rmax((time.perf_counter() - t0) / K, cdev)
        nparam = sum(p.numel() for p in ts.params())
        line = dict(step_ms=step_s * 1e3, geometry_ms=geo, cnn_ms=cnn, collective_ms=coll_s * 1e3, collective_exposed_ms=max(0.0, (step_s - local_s) * 1e3),
                    step_without_exchange_ms=local_s * 1e3, collectives_per_step=out["collectives"], grad_bytes=4 * nparam, last_loss=out["loss"],
                    accepted=out["accepted"], refine_steps_done=out["ref_steps"])
        value = N * world / step_s
        dev_name = torch.cuda.get_device_name(di)
        ts.engine.close()
    if rank == 0:
        out = {"metric": "hypotheses scored/sec (end-to-end training step: 40x40 sub-sampled coord map per frame, BASELINE configs[4])", "value": value, "unit": "hyp/s",
               "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": line["step_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[4]: end-to-end training step, ONE frame per GPU and step (40x40 sub-sampled map, %d hypotheses), "
                                      "scene-coordinate CNN + score CNN (reference architectures, random weights), geometry forward + backward, gradient "
                                      "exchange over %s" % (N, "RCCL" if backend == "nccl" else backend),
                          "frames_per_step_all_ranks": world, "frames_per_s": world / (line["step_ms"] * 1e-3), "ranks_joined": world, "backend": backend,
                          "reduce_mode": args.reduce_mode, "device": dev_name,
                          "parallelism": "data parallel over %d GPU(s): one image per rank, CNN gradients averaged (buckets launched from autograd hooks, "
                                         "waited for at the optimizer step)%s" % (world, ("; " + BACKEND_NOTE) if BACKEND_NOTE else "")},
               "train_step": line}
        if args.dry_run:
            out["dry_run"] = True
        print(json.dumps(out), flush=True)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torchrun environment: this process becomes the launcher of the N ranks
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv))

    rank, local_rank, world, backend, dist = init_distributed(args)
    if args.dry_run:
        run_dry(args, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "config5":
        run_config5(args, rank, local_rank, world, backend, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import torch
    import dsac_amd
    from dsac_amd import synth
    from dsac_amd import dist as ddist

    distributed = world > 1
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    local_rank = local_rank % ndev  # only reachable with a non-RCCL backend (several ranks exercising one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the tiny timing tensors are reduced

    N, H, W = args.hyps, args.height, args.width
    P = H * W
    K, Wm = args.steps, args.warmup
    config3 = args.workload == "config3"

    # frames: seed 1305 + rank (the reference's ThreadRand seed), uploaded before timing
    B = max(1, args.frames_per_step)
    if config3:
        mine = ddist.shard_images(CONFIG3_IMAGES, rank, world)  # image i -> rank i mod world
        B = min(B, max(1, len(mine)))
        if N % 128 != 0:
            raise SystemExit("--workload config3 needs --hyps to be a multiple of 128")
    batched = B > 1 or config3
    if batched and not config3 and (args.kernel_only or args.separate_calls or args.k2_mode != "both" or args.overlap == "stages" or N % 128 != 0):
        B, batched = 1, False  # those modes time single frames (K2-only runs, odd hypothesis counts)
    pipelined = (args.overlap == "pipeline" and not args.kernel_only and args.k2_mode == "both" and not args.separate_calls and not config3)
    fr = synth.chess_like_frame(H, W, seed=1305 + rank)
    if config3:
        xyz_batches = [torch.from_numpy(fr["xyz"]).to(dev)]  # the engine's first frame; the runner below owns the images of the workload
    elif batched:
        def make_batch(tag):
            frs = [synth.chess_like_frame(H, W, seed=1305 + world * 1000 * (tag + 1) + rank * B + f) for f in range(B)]
            return torch.from_numpy(np.ascontiguousarray(np.stack([f_["xyz"] for f_ in frs]))).to(dev)
        xyz_batches = [make_batch(0)] + ([make_batch(1)] if pipelined else [])  # the pipeline alternates between two batches of frames
    else:
        xyz_batches = [torch.from_numpy(fr["xyz"]).to(dev)]
        if pipelined:
            xyz_batches.append(torch.from_numpy(synth.chess_like_frame(H, W, seed=5000 + rank)["xyz"]).to(dev))
    n_ctx = 1 if (pipelined or config3) else max(1, args.streams)
    n_buf = 2 if pipelined else n_ctx
    stride = event_stride_for(K, args.event_stride)
    engines, bufs = [], []

    def set_frames_of(eng, x):
        if batched and not config3:
            eng.set_frames(x, None, H, W, fr["cam"], borrow=True)
        else:
            eng.set_frame(x, None, H, W, fr["cam"], borrow=True)  # implicit full-resolution pixel grid

    for i in range(n_ctx):
        st = torch.cuda.Stream(device=dev)
        eng = dsac_amd.Engine(local_rank, stream=st)
        eng.set_option("k2_flags", K2_FORMS[args.k2_form][0])
        eng.set_option("k2_exact_auto", K2_FORMS[args.k2_form][1])
        SOFT_ONLY_FORM[0] = args.k2_form
        set_frames_of(eng, xyz_batches[0])
        eng.profile_enable(stride > 0, stride=max(1, stride))
        engines.append((eng, st))
    NB = N * B  # hypotheses per launch and context
    # The default step hides its score tail: dsac_set_option("pi_defer_tail", 2) -- the reduction of the per-tile sums and K3 of step i run on the engine's
    # tail stream beside K1 of step i + 1 (include/dsac_hip.h, dsac_score_hypotheses_frames).  Consecutive steps then write alternating result arrays
    # (one error-image buffer: only K2 touches it).  DSAC_BENCH_NO_DEFER=1: everything in stream order, as until round 4 (the A/B).
    defer_tail = bool(batched and not config3 and not pipelined and n_ctx == 1 and not os.environ.get("DSAC_BENCH_NO_DEFER"))
    if defer_tail:
        n_buf = 2
    for i in range(0 if config3 else n_buf):
        bufs.append(dict(
            poses=torch.zeros(NB, 6, dtype=torch.float64, device=dev), sets=torch.zeros(NB, 4, dtype=torch.int32, device=dev),
            ok=torch.zeros(NB, dtype=torch.uint8, device=dev),
            err=bufs[0]["err"] if (defer_tail and i > 0) else torch.empty(NB, P, dtype=torch.float32, device=dev),
            soft=torch.zeros(NB, dtype=torch.float64, device=dev), w=torch.zeros(NB, dtype=torch.float64, device=dev),
            ent=torch.zeros(B, dtype=torch.float64, device=dev), avg=torch.zeros(B, 6, dtype=torch.float64, device=dev)))
    for b in bufs:
        b["err"].zero_()  # first touch of the 2.5 GB of error images happens here, not in a timed launch
    defer_on = [False]

    def set_defer(on):
        defer_on[0] = bool(on and defer_tail)
        engines[0][0].set_option("pi_defer_tail", 2 if defer_on[0] else 0)  # a change of mode orders the stream behind a tail in flight
    set_defer(True)
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
    runner = None
    if config3:
        # per image: refined pose (6) + loss, rotErr, tErr, correct (4) + softmax weights (N) -- what core/test_ransac_softam.cpp:129-263 logs per image.
        # dsac_amd.shard.ShardRunner: batches through dsac_process_images, the refinement tail of a batch under the next batch -- across step boundaries
        # too --, the gather of a step launched on a side stream at the top of the next step and consumed one step later, no host synchronisation.
        from dsac_amd.shard import ShardRunner
        # 2: refinement AND score tail of a batch under the next batch (dsac_hip.h "pi_defer_tail"); DSAC_BENCH_DEFER_MODE=1 / DSAC_BENCH_NO_DEFER for the A/B
        DEFER_MODE = 0 if os.environ.get("DSAC_BENCH_NO_DEFER") else int(os.environ.get("DSAC_BENCH_DEFER_MODE", "2"))
        perm3 = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)

        def frames_of(i):
            return synth.chess_like_frame(H, W, seed=1305 + i)["xyz"]  # seeds 1305 + i
        def make_runner():
            return ShardRunner(engines[0][0], engines[0][1], dev, frames_of, CONFIG3_IMAGES, rank, world, N, H, W, fr["cam"], perm3, batch=B,
                               defer=DEFER_MODE)
        if not os.environ.get("DSAC_BENCH_EM_FIRST"):
            runner = make_runner()
    torch.cuda.synchronize(dev)

    staged = (n_ctx == 2 and args.overlap == "stages" and not args.kernel_only)
    ev_sampled = [torch.cuda.Event() for _ in range(2)]
    ev_scored = [None, None]

    def seed_of(i):
        return 1305 + 7919 * i + rank

    def step_staged(i):
        # two-stage software pipeline over double-buffered pose sets: K1 of frame i+1 runs under K2/K3 of frame i
        (engA, stA), (engB, stB) = engines
        k = i & 1
        b = bufs[k]
        if ev_scored[k] is not None:
            stA.wait_event(ev_scored[k])       # the scoring stage has finished reading this buffer
        engA.sample(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, out=(b["poses"], b["sets"], b["ok"]))
        ev_sampled[k].record(stA)
        stB.wait_event(ev_sampled[k])
        engB.reproject(b["poses"], N=N, clamp=100.0, err=b["err"], soft=b["soft"], tau=10.0, beta=0.5)
        engB.softMax(b["soft"], 0.1, b["poses"], N=N, out=(b["w"], b["ent"], b["avg"]))
        ev_scored[k] = torch.cuda.Event()
        ev_scored[k].record(stB)

    def step_pipelined(i):
        # software pipeline inside ONE context: K1 of step i+1 (aux stream, on the OTHER batch of frames) under K2/K3 of step i
        eng, _ = engines[0]
        k = i & 1
        nb = bufs[1 - k]
        set_frames_of(eng, xyz_batches[(i + 1) & 1])  # borrowed frames: the slot remembers the frame it was sampled from
        eng.sampleAhead(1 - k, NB, seed_of(i + 1), nb["poses"], nb["sets"], nb["ok"], thr=10.0, max_tries=1 << 16)
        b = bufs[k]
        eng.scoreSampled(k, b["poses"], b["soft"], b["w"], ent=b["ent"], avg=b["avg"], err=err_shared, clamp=100.0, tau=10.0, beta=0.5, scale=0.1)

    def step_config3(i):
        runner.step(i)  # enqueues only; results are collected by runner.drain() (inside the timed region, once)

    def step(i):
        if config3:
            return step_config3(i)
        if pipelined:
            return step_pipelined(i)
        if staged:
            return step_staged(i)
        eng, _ = engines[i % n_ctx]
        b = bufs[(i & 1) if defer_on[0] else (i % n_ctx)]
        if batched:
            # three launches for B frames: K1 over B*N waves, K2 over B*N error images, K3 with one workgroup per frame
            eng.scoreHypothesesFrames(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1,
                                      err=b["err"], out=(b["poses"], b["sets"], b["ok"], b["soft"], b["w"], b["ent"], b["avg"]))
            return
        if not args.kernel_only and args.k2_mode == "both" and not args.separate_calls:
            # one C-ABI call: K1 (+ staged pose records) -> K2 -> soft reduce -> K3
            eng.scoreHypotheses(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1,
                                err=b["err"], out=(b["poses"], b["sets"], b["ok"], b["soft"], b["w"], b["ent"], b["avg"]))
            return
        if not args.kernel_only:
            eng.sample(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, out=(b["poses"], b["sets"], b["ok"]))
        eng.reproject(b["poses"], N=N, clamp=100.0, err=b["err"] if args.k2_mode != "soft" else None,
                      soft=b["soft"] if args.k2_mode != "err" else None, tau=10.0, beta=0.5)
        if not args.kernel_only:
            eng.softMax(b["soft"], 0.1, b["poses"], N=N, out=(b["w"], b["ent"], b["avg"]))

    def sync_all():
        for eng, _ in engines:
            eng.synchronize()
        torch.cuda.synchronize(dev)

    def run_emulation():
        """One-GPU emulation of rank r of W (--emulate-world): exactly that rank's images (64 / W per step), its own buffers and launch sequence, the
        all-gather replaced by the copy of its own part.  one_gpu_ms / (W * per_rank_ms) is the strong-scaling efficiency the code path allows when every
        rank has its own GPU (the collective moves 17 KB per rank on a side stream and is not on the critical path)."""
        Wem, rem = args.emulate_world, args.emulate_rank % args.emulate_world
        EM_BASE = int(os.environ.get("DSAC_BENCH_EM_BASE", "100000"))  # first step index of the emulated rank (its seeds: 1305 + 64 * step + image)
        eng0, st0 = engines[0]
        eng0.profile_read(0, reset=True)
        em = ShardRunner(eng0, st0, dev, frames_of, CONFIG3_IMAGES, rem, Wem, N, H, W, fr["cam"], perm3, batch=B, emulate=True,
                         defer=DEFER_MODE,
                         err_buffer=runner.err if (runner is not None and os.environ.get("DSAC_BENCH_EM_SHARE_ERR")) else None)
        # settle like the main run does (--prewarm-ms): the runner's set-up above left the GPU idle for about a second (synthetic frames are made on the
        # host), and 45 steps of 0.5 ms straight out of an idle GPU are timed at a ramping clock (measured: K2 480 us instead of 440)
        t_pre, n_pre_em = time.perf_counter(), 0
        while n_pre_em < max(5, Wm) or (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            em.step(EM_BASE + (n_pre_em % 90))
            n_pre_em += 1
            if n_pre_em % 16 == 0:
                sync_all()
        em.drain()
        sync_all()
        eng0.profile_read(0, reset=True)
        Kem = max(K, 40)
        for key in em.host_us:
            em.host_us[key] = 0.0
        te = time.perf_counter()
        for i in range(Kem):
            em.step(EM_BASE + 100 + i)
        host_s = (time.perf_counter() - te) / Kem  # what the host needs to enqueue a step (it never waits inside one)
        lem = em.drain()
        sync_all()
        per_rank_s = (time.perf_counter() - te) / Kem
        ms_e, n_e = eng0.profile_read(0, reset=True)
        res = {"world": Wem, "rank": rem, "images_per_rank_step": len(em.mine), "batches_per_rank_step": len(em.batches), "steps": Kem,
               "per_rank_ms": per_rank_s * 1e3, "host_enqueue_ms_per_step": host_s * 1e3,
               "host_enqueue_us_by_phase": {k_: v_ / Kem * 1e6 for k_, v_ in em.host_us.items()},
               "k2_us_per_launch": ms_e / max(1, n_e) * 1e3, "rows_finite": bool(torch.isfinite(lem[em.mine]).all()),
               "measured": "before the 64-image run" if os.environ.get("DSAC_BENCH_EM_FIRST") else "after the 64-image run, same process and engine",
               "note": "one GPU runs exactly rank %d's share of %d ranks (its images, buffers, launch sequence, deferred tail, one-step-late exchange; "
                       "the all-gather of 17 KB per rank replaced by the copy of its own part)" % (rem, Wem)}
        em.close()
        for key, v in (("seed_stride", world), ("pi_defer_tail", DEFER_MODE)):
            eng0.set_option(key, v)
        return res

    em_early = None
    if config3 and os.environ.get("DSAC_BENCH_EM_FIRST"):
        # experiment: the emulated rank BEFORE anything of the 64-image run exists (its runner, its 5 GB of error images)
        if args.emulate_world > 1 and world == 1:
            em_early = run_emulation()
        runner = make_runner()

    if pipelined:
        err_shared = bufs[0]["err"]  # the scoring stage is serial: one error-image buffer
        engines[0][0].sampleAhead(0, NB, seed_of(0), bufs[0]["poses"], bufs[0]["sets"], bufs[0]["ok"], thr=10.0, max_tries=1 << 16)
    # untimed settling phase (not steps): the same work until --prewarm-ms have passed, so that clocks, TLBs and the power governor are in
    # their sustained state when the warm-up steps start
    ctr = 0  # steps are numbered consecutively across pre-warm, warm-up and timed region (the pipelined mode's slots alternate strictly)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        step(ctr)
        ctr += 1
        if ctr % 8 == 0:
            sync_all()
    n_pre = ctr
    for i in range(Wm):
        step(ctr)
        ctr += 1
    if config3 and runner.steps_done:
        runner.drain()
    sync_all()
    for eng, _ in engines:
        eng.profile_read(0, reset=True)

    if distributed:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    last = None
    for i in range(K):
        last = step(ctr)
        ctr += 1
    if config3:
        last = runner.drain()  # the last step's refinement tail and gather: part of the job, inside the timed region
    sync_all()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    k2_ms, k2_n = 0.0, 0
    for eng, _ in engines:
        ms, n = eng.profile_read(0, reset=True)
        k2_ms += ms
        k2_n += n
    # the same K steps again, a few times: the spread of the step time on this box (the timed region above is ONE sample of K steps)
    repeats = None
    if rank == 0 and not config3 and not distributed:
        reps = []
        for _ in range(5):
            sync_all()
            tr = time.perf_counter()
            for i in range(K):
                step(ctr)
                ctr += 1
            sync_all()
            reps.append((time.perf_counter() - tr) / K * 1e3)
        repeats = {"n": len(reps), "steps_each": K, "ms_per_step_min": min(reps), "ms_per_step_max": max(reps), "ms_per_step_median": sorted(reps)[len(reps) // 2],
                   "note": "the timed region re-run 5 times after the line's own measurement, same process (box-to-box the same binary differs by several per cent)"}
        for eng, _ in engines:
            eng.profile_read(0, reset=True)
    if not config3:
        set_defer(False)  # the side measurements below reuse one set of result arrays: stream order
    if config3:
        ok_frac = float(runner.scratch[0]["ok"][:N * len(runner.batches[-1])].float().mean().item()) if runner.batches else 1.0  # a rank may own no image
        wsum = float(last[0, 10:].sum().item()) if last is not None else 0.0
    else:
        ok_frac = float(bufs[0]["ok"].float().mean().item()) if not args.kernel_only else 1.0
        wsum = float(bufs[0]["w"][:N].sum().item()) if not args.kernel_only else 1.0

    # The ceiling of K2's own write pattern ON THIS BOX: the same launches with k2_flags bit 1 (the kernel issues its store schedule only, no
    # arithmetic; include/dsac_hip.h).  The pool's boxes differ by several per cent for the same binary; this number moves with them.
    store_only_us = None
    if rank == 0 and not config3 and args.k2_mode != "soft" and not args.separate_calls:
        try:
            for eng, _ in engines:
                eng.set_option("k2_flags", K2_FORMS[args.k2_form][0] | 2)
                eng.profile_read(0, reset=True)
            for i in range(6):
                step(ctr)
                ctr += 1
            sync_all()
            so_ms, so_n = 0.0, 0
            for eng, _ in engines:
                ms, n = eng.profile_read(0, reset=True)
                so_ms += ms
                so_n += n
            store_only_us = so_ms / max(1, so_n) * 1e3 if so_n else None
        finally:
            for eng, _ in engines:
                eng.set_option("k2_flags", K2_FORMS[args.k2_form][0])
            step(ctr)  # leave real results in the buffers
            ctr += 1
            sync_all()
            for eng, _ in engines:
                eng.profile_read(0, reset=True)
    # the OTHER arithmetic forms of K2 at the same launch shape, inside the driver's own line -- the form `value` is measured on and the forms that
    # restate more of the reference's double-precision projection (core/cnn_softam.h:319-362), each with the tolerance it is tested at (`tolerance` below)
    k2_forms = None
    if rank == 0 and not config3 and args.k2_mode != "soft" and not args.separate_calls and not args.kernel_only and not args.no_k2_forms:
        k2_forms = {}
        for name, (flags, auto) in K2_FORMS.items():
            if name == args.k2_form:
                continue
            try:
                for eng, _ in engines:
                    eng.set_option("k2_flags", flags)
                    eng.set_option("k2_exact_auto", auto)
                for i in range(3):
                    step(ctr)
                    ctr += 1
                sync_all()
                for eng, _ in engines:
                    eng.profile_read(0, reset=True)
                tf = time.perf_counter()
                nf = 8
                for i in range(nf):
                    step(ctr)
                    ctr += 1
                sync_all()
                dt = (time.perf_counter() - tf) / nf
                f_ms, f_n = 0.0, 0
                for eng, _ in engines:
                    ms, n = eng.profile_read(0, reset=True)
                    f_ms += ms
                    f_n += n
                if f_n:
                    us = f_ms / f_n * 1e3
                    ab = B * algorithmic_bytes_k2(N, P, explicit_uv=False, write_err=True)
                    k2_forms[name] = {"avg_launch_us": us, "achieved": ab / us / 1e3, "unit": "GB/s", "frac": ab / us / 1e3 / HBM_PEAK_GBS,
                                      "per_image_hyp_s": N * B / dt, "ms_per_step": dt * 1e3, "launches_timed": f_n}
            finally:
                for eng, _ in engines:
                    eng.set_option("k2_flags", K2_FORMS[args.k2_form][0])
                    eng.set_option("k2_exact_auto", K2_FORMS[args.k2_form][1])
        step(ctr)  # leave the line's own form's results in the buffers
        ctr += 1
        sync_all()
        for eng, _ in engines:
            eng.profile_read(0, reset=True)
    # SURVEY.md 8(d) secondary line: the same launches WITHOUT the error-image output (fused soft-inlier mode): K2 is then VALU-bound
    soft_only = None
    if rank == 0 and batched and not config3 and not pipelined and args.k2_mode == "both":
        eng, _ = engines[0]
        b = bufs[0]
        eng.profile_read(0, reset=True)
        for i in range(12):
            eng.scoreHypothesesFrames(N, seed=seed_of(ctr + i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=None,
                                      out=(b["poses"], b["sets"], b["ok"], b["soft"], b["w"], b["ent"], b["avg"]))
            if i == 1:
                eng.synchronize()
                eng.profile_read(0, reset=True)  # the first two launches settle
        eng.synchronize()
        ms_s, n_s = eng.profile_read(0, reset=True)
        if n_s:
            soft_only = soft_only_roofline(N * B, P, ms_s / n_s * 1e-3, n_s)
        step(ctr)  # leave real error images in the buffers
        ctr += 1
        sync_all()
        for e_, _ in engines:
            e_.profile_read(0, reset=True)
    emulation = None
    if config3 and rank == 0:
        ws = last[:, 10:].sum(1)
        assert bool(((ws - 1.0).abs() < 1e-9).all()), "config3: gathered softmax weights do not sum to 1 for every image"
        assert bool((last[:, 6] > 0).all()) and bool(torch.isfinite(last[:, :10]).all()), "config3: gathered refined poses / losses are not finite"
    if config3 and args.emulate_world > 1 and world == 1:
        em_raw = em_early if em_early is not None else run_emulation()
        one_gpu_s = elapsed / K
        Wem = em_raw.pop("world")
        emulation = dict(world=Wem, **em_raw)
        emulation.update({"one_gpu_ms": one_gpu_s * 1e3, "ideal_per_rank_ms": one_gpu_s * 1e3 / Wem,
                          "predicted_speedup": one_gpu_s / (em_raw["per_rank_ms"] * 1e-3), "predicted_efficiency": one_gpu_s / (em_raw["per_rank_ms"] * 1e-3) / Wem})

    # literal configs[1]: ONE frame per step on the same context (fused call), its own K2 timing
    single = None
    if not args.no_single_frame and not args.kernel_only and not config3 and args.k2_mode == "both" and rank == 0:
        eng, _ = engines[0]
        b = bufs[0]
        x1 = xyz_batches[0][0] if batched else xyz_batches[0]
        eng.set_frame(x1, None, H, W, fr["cam"], borrow=True)
        eng.profile_enable(True, stride=1)

        def one(i):
            eng.scoreHypotheses(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=b["err"][:N],
                                out=(b["poses"][:N], b["sets"][:N], b["ok"][:N], b["soft"][:N], b["w"][:N], b["ent"][:1], b["avg"][0]))
        for i in range(20):
            one(ctr + i)
        eng.synchronize()
        eng.profile_read(0, reset=True)
        n1 = 100
        t1 = time.perf_counter()
        for i in range(n1):
            one(ctr + 20 + i)
        eng.synchronize()
        e1 = time.perf_counter() - t1
        ms1, c1 = eng.profile_read(0, reset=True)
        # the frame time itself without the two event records per step that time K2
        eng.profile_enable(False)
        t1 = time.perf_counter()
        for i in range(n1):
            one(ctr + 20 + n1 + i)
        eng.synchronize()
        e1 = min(e1, time.perf_counter() - t1)
        eng.profile_enable(True, stride=1)
        # ... and the same loop with the score tail of frame i beside K1 of frame i + 1 ("pi_defer_tail" 2, alternating result arrays)
        b2 = bufs[1] if len(bufs) > 1 else {k_: (torch.zeros_like(v_) if k_ != "err" else v_) for k_, v_ in b.items()}

        def one_d(i):
            bb = b2 if (i & 1) else b
            eng.scoreHypotheses(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=b["err"][:N],
                                out=(bb["poses"][:N], bb["sets"][:N], bb["ok"][:N], bb["soft"][:N], bb["w"][:N], bb["ent"][:1], bb["avg"][0]))
        eng.profile_enable(False)
        eng.set_option("pi_defer_tail", 2)
        for i in range(20):
            one_d(i)
        eng.synchronize()
        t1 = time.perf_counter()
        for i in range(n1):
            one_d(20 + i)
        eng.synchronize()
        e1d = time.perf_counter() - t1
        eng.set_option("pi_defer_tail", 0)
        eng.profile_enable(True, stride=1)
        k2_1 = ms1 / max(1, c1) * 1e-3
        ab1 = algorithmic_bytes_k2(N, P, explicit_uv=False)
        single = {"workload": "BASELINE.json configs[1] literally: ONE %dx%d frame x %d hypotheses per step (K1 -> K2 -> K3, one context, no overlap)" % (W, H, N),
                  "value": N * n1 / e1, "unit": "hyp/s", "steps": n1, "us_per_frame": e1 / n1 * 1e6,
                  "score_tail_under_the_next_frame": {"value": N * n1 / e1d, "unit": "hyp/s", "us_per_frame": e1d / n1 * 1e6,
                                                      "what": "the same loop with dsac_set_option(pi_defer_tail, 2): reduction + K3 of frame i beside K1 of frame i + 1"},
                  "roofline": {"kernel": "k_reproject (K2)", "achieved": ab1 / k2_1 / 1e9 if k2_1 > 0 else 0.0, "unit": "GB/s",
                               "frac": (ab1 / k2_1 / 1e9 / HBM_PEAK_GBS) if k2_1 > 0 else 0.0, "avg_launch_us": k2_1 * 1e6, "launches_timed": c1,
                               "algorithmic_bytes_per_launch": ab1}}

    # BASELINE.json configs[0] / SURVEY.md 8(d) config 1 in the reference-exact mode: P = 40 x 40 sub-sampled map, int16-quantised
    # coordinates, 256 hypotheses, one frame per step (K1 -> K2 -> K3); 32 such frames per step as well (what a GPU is fed with)
    refsize = None
    if not args.no_single_frame and not args.kernel_only and not config3 and args.k2_mode == "both" and rank == 0:
        eng, _ = engines[0]
        f40 = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
        x40 = torch.from_numpy(f40["xyz"]).to(dev)
        u40 = torch.from_numpy(f40["uv"]).to(dev)
        b = bufs[0]
        eng.profile_enable(False)  # no K2 timing from here on: no event records in these launch chains
        refsize = {"workload": "reference-exact size: 40x40 stratified sub-sample, int16-quantised coordinates, %d hypotheses (core/lua_calls.h:33, core/types.h:43)" % N}
        for nf in (1, 32):
            if nf == 1:
                eng.set_frame(x40, u40, 40, 40, f40["cam"], borrow=True)
            else:
                xs = torch.from_numpy(np.ascontiguousarray(np.stack([synth.chess_like_frame(40, 40, seed=1305 + k, quantise_int16=True)["xyz"] for k in range(nf)]))).to(dev)
                eng.set_frames(xs, u40, 40, 40, f40["cam"], borrow=True)
            nn = N * nf
            if nf > 1 and N % 128 != 0:
                continue
            o = (torch.zeros(nn, 6, dtype=torch.float64, device=dev), torch.zeros(nn, 4, dtype=torch.int32, device=dev), torch.zeros(nn, dtype=torch.uint8, device=dev),
                 torch.zeros(nn, dtype=torch.float64, device=dev), torch.zeros(nn, dtype=torch.float64, device=dev), torch.zeros(nf, dtype=torch.float64, device=dev),
                 torch.zeros(nf, 6, dtype=torch.float64, device=dev))
            e40 = torch.empty(nn, 1600, dtype=torch.float32, device=dev)

            def one40(i):
                if nf == 1:
                    eng.scoreHypotheses(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=e40,
                                        out=(o[0], o[1], o[2], o[3], o[4], o[5][:1], o[6][0]))
                else:
                    eng.scoreHypothesesFrames(N, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=e40, out=o)
            for i in range(20):
                one40(i)
            eng.synchronize()
            n40 = 200
            t40 = time.perf_counter()
            for i in range(n40):
                one40(20 + i)
            eng.synchronize()
            e40s = time.perf_counter() - t40
            refsize["frames_per_step_%d" % nf] = {"value": nn * n40 / e40s, "unit": "hyp/s", "us_per_frame": e40s / n40 / nf * 1e6}
        eng.profile_read(0, reset=True)

    # the reference's test-time unit of work (test_ransac_softam.cpp:97-157): processImage of ONE image -- K1 -> K2 -> K3, then the 8-step
    # inlier refinement of the soft-argmax pose (K6) and the pose loss (K7) -- device-resident, at full and at reference size
    procimg = None
    if not args.no_single_frame and not args.kernel_only and not config3 and args.k2_mode == "both" and rank == 0:
        from dsac_amd.capi import lib as _lib, ptr as _ptr, check as _check
        eng, _ = engines[0]
        b = bufs[0]
        eng.profile_enable(False)
        procimg = {"workload": "processImage of one image (core/cnn_softam.h:960-1179 with the soft-inlier score): sample + P3P, error images, softmax / "
                               "soft-argmax, 8 refinement steps, pose loss; %d hypotheses, everything resident in HBM" % N}
        f40 = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
        for (hh, ww, xs, us, cam_) in ((H, W, xyz_batches[0][0] if batched else xyz_batches[0], None, fr["cam"]),
                                       (40, 40, torch.from_numpy(f40["xyz"]).to(dev), torch.from_numpy(f40["uv"]).to(dev), f40["cam"])):
            pp = hh * ww
            eng.set_frame(xs, us, hh, ww, cam_, borrow=True)
            perm_d = torch.from_numpy(synth.fast_permutations(pp, 8)).to(dev)
            gt_d = torch.zeros(6, dtype=torch.float64, device=dev)
            ref_d, out4_d, sd_d = torch.zeros(6, dtype=torch.float64, device=dev), torch.zeros(4, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
            errp = b["err"][:N] if pp == P else torch.empty(N, pp, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

            one_out = dict(hyps=b["poses"][:N], sampledPoints=b["sets"][:N], ok=b["ok"][:N], scores=b["soft"][:N], sfScores=b["w"][:N], sfEntropy=b["ent"][:1],
                           avgHyp=b["avg"][:1], refAvgHyp=ref_d.view(1, 6), refSteps=sd_d, out4=out4_d.view(1, 4))
            gt_row = gt_d.view(1, 6)

            def proc(i):
                # ONE call, in stream order (dsac_process_images on a single frame: K1, K2, the score tail in one launch, K6 with the loss at its end) --
                # what Frame::processImage of the C++ surface issues; until round 4 this leg timed the same stages as three calls (K7 its own launch)
                eng.processImages(N, perm_d, gt_jp6=gt_row, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=errp, out=one_out)
            for i in range(10):
                proc(i)
            eng.synchronize()
            npi = 100
            tp = time.perf_counter()
            for i in range(npi):
                proc(10 + i)
            eng.synchronize()
            procimg["%dx%d" % (ww, hh)] = {"us_per_image": (time.perf_counter() - tp) / npi * 1e6, "images": npi, "refine_steps_done": int(sd_d.item())}
            # the same unit as a STREAM of images (the loop over images of test_ransac_softam.cpp:97): one dsac_process_images chain per image, the
            # refinement tail of image i under sampling / scoring of image i + 1 (dsac_hip.h "pi_defer_tail" = 1; = 2 the score tail as well).  The
            # latency of ONE image stays the in-order figure above; this is what a loop over images pays per image
            f64_ = dict(dtype=torch.float64, device=dev)
            two = [dict(hyps=torch.zeros(N, 6, **f64_), sampledPoints=torch.zeros(N, 4, dtype=torch.int32, device=dev), ok=torch.zeros(N, dtype=torch.uint8, device=dev),
                        scores=torch.zeros(N, **f64_), sfScores=torch.zeros(N, **f64_), sfEntropy=torch.zeros(1, **f64_), avgHyp=torch.zeros(1, 6, **f64_),
                        refAvgHyp=torch.zeros(1, 6, **f64_), refSteps=torch.zeros(1, dtype=torch.int32, device=dev), out4=torch.zeros(1, 4, **f64_)) for _ in (0, 1)]
            gt1 = torch.zeros(1, 6, **f64_)
            torch.cuda.synchronize()
            # arguments bound once (Engine.bindProcessImages: no marshalling of two dozen buffers per image).  In this process the 640x480 loop reads ~90 us
            # per image with both tails deferred against 81 from the C++ program -- not host time (the 40x40 loop reaches 57 us here)
            bound = [eng.bindProcessImages(N, perm_d, two[k_], gt_jp6=gt1, thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=errp) for k_ in (0, 1)]
            eng.set_option("device_args", 1)  # every bound argument lives in HBM: no pointer query per argument either
            for mode_ in (1, 2):
                eng.set_option("pi_defer_tail", mode_)
                for i in range(10):
                    bound[i & 1](seed_of(i))
                eng.synchronize()
                tp = time.perf_counter()
                for i in range(npi):
                    bound[i & 1](seed_of(10 + i))
                eng.synchronize()
                procimg["%dx%d_stream_of_images_%s_under_the_next_image" % (ww, hh, "refinement" if mode_ == 1 else "score_and_refinement")] = {
                    "us_per_image": (time.perf_counter() - tp) / npi * 1e6, "images": npi, "refine_steps_done": int(two[(npi - 1) & 1]["refSteps"].item())}
            eng.set_option("pi_defer_tail", 0)
            eng.set_option("device_args", 0)
        if batched:
            # the same unit for the %d frames of a step in ONE call (dsac_process_images: one launch per stage, K6 one wave per frame)
            Bf = B
            eng.set_frames(xyz_batches[0], None, H, W, fr["cam"], borrow=True)
            permB = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
            gtB = torch.zeros(Bf, 6, dtype=torch.float64, device=dev)
            refB, sdB, o4B = torch.zeros(Bf, 6, dtype=torch.float64, device=dev), torch.zeros(Bf, dtype=torch.int32, device=dev), torch.zeros(Bf, 4, dtype=torch.float64, device=dev)
            outB = dict(hyps=b["poses"], sampledPoints=b["sets"], ok=b["ok"], scores=b["soft"], sfScores=b["w"], sfEntropy=b["ent"], avgHyp=b["avg"],
                        refAvgHyp=refB, refSteps=sdB, out4=o4B)

            def procB(i):
                eng.processImages(N, permB, gt_jp6=gtB, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=b["err"], out=outB)
            outB2 = {k_: torch.zeros_like(v_) for k_, v_ in outB.items()}  # defer = 2: consecutive calls write different arrays (dsac_hip.h)

            def procB2(i):
                eng.processImages(N, permB, gt_jp6=gtB, seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, scale=0.1, err=b["err"],
                                  out=outB2 if (i & 1) else outB)
            for key, defer in (("%dx%d_batch_of_%d" % (W, H, Bf), 0), ("%dx%d_batch_of_%d_refinement_under_the_next_batch" % (W, H, Bf), 1),
                               ("%dx%d_batch_of_%d_score_and_refinement_under_the_next_batch" % (W, H, Bf), 2)):
                # defer = 1: dsac_set_option("pi_defer_tail"): K6 / K7 of a batch run on their own stream under K1 / K2 of the next one; 2: the score
                # reduction and K3 too (K1 of the next batch follows K2 directly)
                eng.set_option("pi_defer_tail", defer)
                procB = procB2 if defer == 2 else procB
                for i in range(5):
                    procB(i)
                eng.synchronize()
                nb_ = 30
                tp = time.perf_counter()
                for i in range(nb_):
                    procB(5 + i)
                eng.synchronize()
                procimg[key] = {"us_per_image": (time.perf_counter() - tp) / nb_ / Bf * 1e6, "images": nb_ * Bf, "refine_steps_done_min": int(sdB.min().item())}
            # the same batch through the score-CNN seam (cnn_softam.h:1066-1078): dsac_process_images_begin leaves the error images in HBM, the scores come
            # from OUTSIDE (here: the soft-inlier sums begin wrote on the side, standing in for a score model's output), dsac_process_images_finish continues
            def procS(i):
                o = outB2 if (i & 1) else outB
                eng.processImagesBegin(N, b["err"], seed=seed_of(i), thr=10.0, max_tries=1 << 16, clamp=100.0, tau=10.0, beta=0.5, soft=o["scores"],
                                       out=(o["hyps"], o["sampledPoints"], o["ok"]))
                eng.processImagesFinish(N, o["scores"], permB, o["hyps"], gt_jp6=gtB, scale=0.1, thr=10.0, out=o)
            for key, defer in (("%dx%d_batch_of_%d_external_scores" % (W, H, Bf), 0),
                               ("%dx%d_batch_of_%d_external_scores_score_and_refinement_under_the_next_batch" % (W, H, Bf), 2)):
                eng.set_option("pi_defer_tail", defer)
                for i in range(5):
                    procS(i)
                eng.synchronize()
                tp = time.perf_counter()
                for i in range(nb_):
                    procS(5 + i)
                eng.synchronize()
                procimg[key] = {"us_per_image": (time.perf_counter() - tp) / nb_ / Bf * 1e6, "images": nb_ * Bf, "refine_steps_done_min": int(sdB.min().item()),
                                "what": "dsac_process_images_begin -> scores from outside (the soft-inlier sums as a stand-in) -> dsac_process_images_finish"}
            eng.set_option("pi_defer_tail", 0)
        eng.profile_read(0, reset=True)

    # north_star's multi-GPU claim in the driver's own command: with --gpus N > 1 the line carries a `strong` object -- configs[3] (64 images fixed) sharded
    # over the N ranks, the result rows exchanged over RCCL inside the timed region -- next to the weak-scaling `value`; on one GPU --emulate-world W fills
    # the same fields from the emulation of one rank's share
    strong = None
    if not config3 and not args.kernel_only and args.k2_mode == "both" and (world > 1 or args.emulate_world > 1):
        for b_ in bufs[1:]:
            b_["err"] = None  # the leg's runners write into the first context's error-image buffer; drop the others
        strong = strong_scaling_leg(args, rank, local_rank, world, backend, dist if distributed else None, emulate_world=args.emulate_world,
                                    emulate_rank=args.emulate_rank, err_buffer=bufs[0]["err"] if (batched and bufs[0]["err"] is not None) else None,
                                    engine=engines[0][0], stream=engines[0][1])
        engines[0][0].profile_read(0, reset=True)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kk = torch.tensor([k2_ms, float(k2_n)], dtype=torch.float64, device=cdev)
        dist.all_reduce(kk, op=dist.ReduceOp.SUM)
        k2_ms, k2_n = float(kk[0].item()), int(kk[1].item())

    if rank == 0:
        if config3:
            total_hyps = CONFIG3_IMAGES * N * K
            frames_per_launch = B
        else:
            total_hyps = N * B * K * world
            frames_per_launch = B
        value = total_hyps / elapsed
        k2_avg_s = (k2_ms / max(1, k2_n)) * 1e-3
        abytes = frames_per_launch * algorithmic_bytes_k2(N, P, explicit_uv=False, write_err=args.k2_mode != "soft")  # one launch carries B frames
        achieved = abytes / k2_avg_s / 1e9 if k2_avg_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the counters of THIS launch shape and of the K2 form that is timed (files from before round 6 carry no form: the fp32 one)
                if tj.get("N") == N * B and tj.get("P") == P and tj.get("frames", 1) == B and tj.get("form", "fast") == args.k2_form:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if config3:
            cfg_name = ("BASELINE.json configs[3]: %d 'chess'-like images x %d hypotheses, %dx%d coord maps, sharded round-robin over %d rank(s) "
                        "(%d images per launch), every image through the whole processImage (K1 sample+P3P, K2, K3, 8 refinement steps K6, loss K7), "
                        "results (refined pose 6 + loss / errors 4 + N weights per image) gathered" % (CONFIG3_IMAGES, N, W, H, world, B))
        elif args.kernel_only:
            cfg_name = ("BASELINE.json configs[2]: %d random poses over a %dx%d coord map, K2 only" % (N, W, H) if N >= 1024 else
                        "K2 only on %d random poses over a %dx%d coord map (BASELINE.json configs[2] style at configs[1]'s hypothesis count)" % (N, W, H))
        else:
            cfg_name = ("BASELINE.json configs[1]: 'chess'-like frame, %d hypotheses, %dx%d coord map%s, K1 sample+P3P -> K2 reproject (error images + "
                        "soft-inlier) -> K3 softmax" % (N, W, H, (" x %d independent frames per step (one launch each for K1, K2, K3)" % B) if batched
                                                         else " (single frame per step)"))
        out = {
            "metric": "hypotheses scored/sec over 640x480 coord map",
            "value": value, "unit": "hyp/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong" if config3 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name,
                       "hypotheses_per_frame": N, "frame": [H, W], "frames_per_step": CONFIG3_IMAGES if config3 else B, "streams_per_gpu": n_ctx,
                       "overlap": ("in-context software pipeline: K1(i+1) || K2,K3(i), alternating frame batches" if pipelined else
                                   "K2 launches serialised across 2 contexts, K1/K3 overlap them" if gated else "sampling stage || scoring stage" if staged
                                   else ("frames round-robin" if n_ctx > 1 else
                                         ("score tail of step i (reduction of the per-tile sums + K3) on the tail stream beside K1 of step i + 1 "
                                          "(pi_defer_tail = 2); K1 -> K2 in stream order" if defer_tail else "none"))),
                       "parallelism": "images sharded over %d GPU(s), no data-path collective%s%s" %
                                      (world, "; results gathered on rank 0" if config3 else "", ("; " + BACKEND_NOTE) if BACKEND_NOTE else ""),
                       "prewarm_steps_untimed": n_pre, "accepted_fraction": ok_frac, "softmax_sum": wsum},
            "roofline": ({"kernel": "k_reproject (K2)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                          "traffic_source": ("constant from the committed PMC pass profiles/k2_traffic.json (rocprofv3 WRITE_SIZE + 2 x FETCH_SIZE of this "
                                             "launch shape), not measured in this run") if traffic is not None else None,
                          "algorithmic_bytes_per_launch": abytes, "avg_launch_us": k2_avg_s * 1e6, "launches_timed": k2_n, "event_stride": stride,
                          "store_schedule_only_us": store_only_us}
                         if args.k2_mode != "soft" else
                         dict(soft_only_roofline(N * frames_per_launch, P, k2_avg_s, k2_n), traffic=None, event_stride=stride,
                              algorithmic_bytes_per_launch=abytes)),
            # kernel-only (K2) and per-image (K1 + K2 + K3) rates reported separately
            "rates": {"per_image_hyp_s": value, "kernel_only_k2_hyp_s": (N * frames_per_launch / k2_avg_s * world) if k2_avg_s > 0 else None,
                      "unit": "hyp/s", "note": "per_image = whole step (K1 sample+P3P, K2, soft reduce, K3); kernel_only = hypotheses per K2 launch / its duration"},
        }
        out["tolerance"] = dict(K2_TOLERANCES, value_measured_on=args.k2_form)
        if k2_forms:
            out["k2_forms"] = k2_forms
        if repeats is not None:
            out["repeats"] = repeats
        if soft_only is not None:
            out["soft_only"] = soft_only
        if single is not None:
            out["single_frame"] = single
        if refsize is not None:
            out["reference_size"] = refsize
        if procimg is not None:
            out["process_image"] = procimg
        if emulation is not None:
            out["emulation"] = emulation
        if strong is not None:
            out["strong"] = strong
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, fr, N, H, W)
        if (not args.no_host_driver and world == 1 and not config3 and not args.kernel_only and args.k2_mode == "both" and not args.no_single_frame
                and N % 128 == 0):
            hd = host_driver_leg(N, H, W, device=local_rank)
            ref = (procimg or {}).get("%dx%d_batch_of_%d_refinement_under_the_next_batch" % (W, H, B)) if batched else None
            if ref and "us_per_image" in hd:
                hd["vs_process_image_python_ctypes"] = hd["us_per_image"] / ref["us_per_image"]  # the same unit of work driven from Python (process_image above)
            out["host_driver"] = hd
        print(json.dumps(out), flush=True)

    for eng, _ in engines:
        eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""CPU-side checks of bench.py's contract: the algorithmic-byte formula of SURVEY.md 8(d), the committed bench line of the
round (profiles/r01_bench_default.json) carrying every field the driver and the judge read, and its internal consistency."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_formula():
    b = _bench()
    P = 640 * 480
    # 1 228 852 B + 3 686 400 / N B per hypothesis at P = 307 200, implicit grid
    assert b.algorithmic_bytes_k2(256, P, explicit_uv=False) == 256 * 1228852 + 3686400
    assert b.algorithmic_bytes_k2(4096, P, explicit_uv=False) == 4096 * 1228852 + 3686400
    assert b.algorithmic_bytes_k2(1, 1600, explicit_uv=True) == 12 * 1600 + 8 * 1600 + 48 + 4 * 1600 + 4
    assert b.algorithmic_bytes_k2(256, P, explicit_uv=False, write_err=False) == 12 * P + 48 * 256 + 4 * 256


def test_soft_only_mode_is_priced_against_the_valu_roof():
    """the fused soft-inlier mode is VALU-bound (36 flop per pair, 12 B per pixel); its roofline object names the fp32 vector
    peak and never an HBM fraction."""
    b = _bench()
    r = b.soft_only_roofline(2048, 640 * 480, 384e-6, 10)  # round 2's measurement: 384 us per 8-frame launch
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == b.VALU_PEAK_TFLOPS == 157.3
    assert r["flop_per_launch"] == 2048 * 640 * 480 * 36 and abs(r["achieved"] - 58.98) < 0.01 and abs(r["frac"] - 0.375) < 1e-3
    assert "GB/s" not in json.dumps(r) and "hbm" not in json.dumps(r).lower()


import pytest


@pytest.mark.parametrize("name", ["r01_bench_default.json", "r02_bench_driver_flags.json", "r02_final_bench_driver_flags.json",
                                  "r03_final_bench_driver_flags.json", "r05_final_bench_driver_flags.json", "r06_final_bench_driver_flags.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    d = json.loads(line)
    r06 = name.startswith("r06")
    if r06:
        # round 6: `value` is measured on the K2 form that holds every stated tolerance (the exact transform) and the line says so; the other arithmetic forms
        # ride along with their launch times; the PMC traffic is that of the timed form
        tol = d["tolerance"]
        assert tol["value_measured_on"] == "exact" and tol["exact"]["cells_above_1e-3_px"] == "0" and tol["exact"]["near_tie_weight_error"] <= tol["stated"]["softmax_weight"]
        assert tol["exact"]["all_cells_max_px"] <= tol["stated"]["residual_px"] < tol["fast"]["all_cells_max_px"]
        kf = d["k2_forms"]
        assert set(kf) == {"fast", "precise"} and kf["fast"]["avg_launch_us"] < d["roofline"]["avg_launch_us"] < kf["precise"]["avg_launch_us"]
        for v in kf.values():
            assert abs(v["frac"] - v["achieved"] / 8000.0) < 1e-9
        assert d["roofline"]["frac"] >= 0.60 and "exact" in json.load(open(os.path.join(ROOT, "profiles", "k2_traffic.json")))["form"]
        assert d["rates"]["per_image_hyp_s"] / d["rates"]["kernel_only_k2_hyp_s"] >= 0.925
        st = json.loads(open(os.path.join(ROOT, "profiles", "r06_final_bench_em8_rank0.json")).read().strip().splitlines()[-1])["strong"]
        _check_strong(st, ranks=1, shards=8)
        assert st["emulated"] is True and 6.0 < st["speedup"] <= 8.0 and "host_enqueue_ms_per_step" in st
        name = "r05x_" + name
    if name.startswith("r05"):
        # round 5: the default step hides its score tail and says so; min / max of five re-runs of the timed region; the seam's cost next to the built-in score;
        # the training round without error images; no `strong` object on one GPU unless asked for (it is in r05_final_bench_em8_rank0.json)
        assert "pi_defer_tail = 2" in d["config"]["overlap"] and "strong" not in d
        rp = d["repeats"]
        assert rp["n"] == 5 and rp["ms_per_step_min"] <= rp["ms_per_step_median"] <= rp["ms_per_step_max"] and abs(rp["ms_per_step_median"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
        pi = d["process_image"]
        ext, own = pi["640x480_batch_of_16_external_scores"]["us_per_image"], pi["640x480_batch_of_16"]["us_per_image"]
        assert abs(ext - own) < 0.08 * own  # within a few per cent of the built-in one
        assert d["host_driver"]["training"]["us_per_frame_without_error_images"] < d["host_driver"]["training"]["us_per_frame"]
        # 0.939-0.942 with the score tail hidden (the first closing run of round 5); K1 then took on OpenCV's arithmetic and alignment (+7 us on the
        # critical path, profiles/r05_k1_cost.txt): 0.931-0.937
        assert d["rates"]["per_image_hyp_s"] / d["rates"]["kernel_only_k2_hyp_s"] >= 0.925
        if not r06:
            st = json.loads(open(os.path.join(ROOT, "profiles", "r05_final_bench_em8_rank0.json")).read().strip().splitlines()[-1])["strong"]
            _check_strong(st, ranks=1, shards=8)
            assert st["emulated"] is True and 6.0 < st["speedup"] <= 8.0
        name = "r03_" + name
    if name.startswith("r03"):
        # round 3: the secondary measurement of SURVEY.md 8(d) rides in the driver's line, priced against the VALU roof; the traffic figure says
        # where it comes from; the batched processImage; the CPU baseline on the cores the container is granted
        so = d["soft_only"]
        assert so["bound"] == "valu" and so["unit"] == "TFLOP/s" and so["peak"] == 157.3 and abs(so["frac"] - so["achieved"] / 157.3) < 1e-9
        assert so["flop_per_launch"] == 16 * 256 * 640 * 480 * 36 and 0.2 < so["frac"] < 1.0
        assert "PMC" in d["roofline"]["traffic_source"] and d["roofline"]["frac"] >= (0.60 if r06 else 0.70)  # round 6 times the exact-transform form
        assert d["process_image"]["640x480_batch_of_16"]["us_per_image"] < 0.5 * d["process_image"]["640x480"]["us_per_image"]
        assert d["cpu_baseline"]["cores"] <= 64 and "quota" in d["cpu_baseline"]["cpus"]
        d = dict(d)  # the shared checks below know the round-2 shape
        name = "r02_" + name
    if name.startswith("r02"):
        # round 2: the driver's own flags, every K2 launch of the timed region timed, the extra SURVEY 8(d) fields
        assert d["steps"] == 20 and d["warmup"] == 5 and d["roofline"]["launches_timed"] == 20 and d["roofline"]["frac"] >= 0.60
        s1 = d["single_frame"]
        assert s1["roofline"]["algorithmic_bytes_per_launch"] == _bench().algorithmic_bytes_k2(256, 640 * 480, explicit_uv=False)
        assert abs(s1["roofline"]["frac"] - s1["roofline"]["achieved"] / 8000.0) < 1e-9 and s1["value"] < d["value"]
        r = d["rates"]
        assert abs(r["per_image_hyp_s"] - d["value"]) < 1e-6 * d["value"] and r["kernel_only_k2_hyp_s"] > r["per_image_hyp_s"]
        assert d["cpu_baseline"]["one_thread"]["cores"] == 1 and d["cpu_baseline"]["one_thread"]["value"] < d["cpu_baseline"]["value"]
    if name.startswith("r02_final"):
        # the closing line of round 2: 16 frames per step, the store-only twin of the same launches, processImage of one image
        assert d["config"]["frames_per_step"] == 16
        so = d["roofline"]["store_schedule_only_us"]
        assert so is not None and abs(so - d["roofline"]["avg_launch_us"]) < 0.1 * d["roofline"]["avg_launch_us"]  # K2 sits on its store schedule
        pi = d["process_image"]
        assert pi["640x480"]["refine_steps_done"] == 8 and pi["40x40"]["refine_steps_done"] == 8
        assert 50.0 < pi["40x40"]["us_per_image"] < pi["640x480"]["us_per_image"] < 1000.0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "hyp/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    B, N = d["config"]["frames_per_step"], d["config"]["hypotheses_per_frame"]
    assert r["algorithmic_bytes_per_launch"] == B * _bench().algorithmic_bytes_k2(N, 640 * 480, explicit_uv=False)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    # whole-job throughput = hypotheses per step / time per step
    assert abs(d["value"] - B * N * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # PMC traffic within a few percent of the algorithmic bytes (no wasted re-reads)
    assert r["traffic"] is not None and 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def _run_bench(*flags, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=e, capture_output=True, text=True, timeout=300)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` without a torchrun environment spawns 2 ranks (gloo here, RCCL on the GPUs), and the line reports the
    number of ranks that joined.  --dry-run replaces the engine; launch, rendezvous, MAX-over-ranks timing and the line are the real code."""
    r = _run_bench("--gpus", "2", "--steps", "3", "--dry-run")
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["frames_per_step_all_ranks"] == 32  # 16 frames per rank
    # round 5: with more than one rank the SAME line carries north_star's strong-scaling claim -- configs[3], 64 images fixed, sharded, the result rows
    # exchanged by the backend inside the timed region -- next to the weak-scaling value
    _check_strong(d["strong"], ranks=2, shards=2)


def test_the_command_the_driver_runs_at_eight_ranks():
    """`bench.py --gpus 8` as the driver launches it on an 8-GPU node -- 8 processes, one rendezvous, the weak line plus the `strong` object
    (configs[3]: 64 images, 8 per rank and step, one gather of 8 x 8 x (10 + N) doubles) -- exercised here with 8 CPU ranks over gloo (--dry-run: the stand-in
    engine; launch, sharding, exchange, max-over-ranks timing and reporting are the real code)."""
    r = _run_bench("--gpus", "8", "--steps", "2", "--dry-run")
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["frames_per_step_all_ranks"] == 8 * 16
    s = d["strong"]
    _check_strong(s, ranks=8, shards=8)
    assert s["ranks_joined"] == 8 and s["images_per_rank_step"] == 8 and s["collective_bytes_per_step"] == 8 * 8 * (10 + 256) * 8 and s["rows_ok"] is True
    assert s["host_enqueue_ms_per_step"] > 0  # the slowest rank's host time per step (the real leg: whether 8 processes on the node's cores are the limiter)


STRONG_FIELDS = ("workload", "ranks_joined", "backend", "steps", "images_per_rank_step", "one_gpu_ms", "per_rank_ms", "speedup", "efficiency",
                 "collective_bytes_per_step", "collective_exposed_us", "rows_ok")


def _check_strong(s, ranks, shards):
    for k in STRONG_FIELDS:
        assert k in s, k
    assert s["ranks_joined"] == ranks and s["images_per_rank_step"] == 64 // shards and s["rows_ok"] is True
    assert s["one_gpu_ms"] > 0 and s["per_rank_ms"] > 0 and abs(s["speedup"] - s["one_gpu_ms"] / s["per_rank_ms"]) < 1e-9 * s["speedup"]
    assert abs(s["efficiency"] - s["speedup"] / shards) < 1e-12
    assert s["collective_bytes_per_step"] == shards * (64 // shards) * (10 + 256) * 8  # ranks x rows per rank x (6 + 4 + N) doubles


def test_one_gpu_emulation_fills_the_same_strong_fields():
    """--gpus 1 --emulate-world 8: the one-GPU box's line carries the same object, flagged as an emulation (a prediction, not a measurement)."""
    r = _run_bench("--gpus", "1", "--steps", "3", "--dry-run", "--emulate-world", "8")
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak"
    _check_strong(d["strong"], ranks=1, shards=8)
    assert d["strong"]["emulated"] is True
    r = _run_bench("--gpus", "1", "--steps", "3", "--dry-run")
    assert "strong" not in _json_line(r.stdout)


def test_config3_shards_64_images_and_gathers_them():
    """BASELINE.json configs[3]: 64 images round-robin over the ranks, 64 x (6 + N) results gathered (dist.shard_images / gather_frame_results);
    the dry run fills row i with the image index and checks on rank 0 that every image arrived in its place."""
    for world in (1, 2, 3):
        r = _run_bench("--gpus", str(world), "--steps", "2", "--workload", "config3", "--dry-run", "--hyps", "128")
        assert r.returncode == 0, r.stderr
        d = _json_line(r.stdout)
        assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["frames_per_step_all_ranks"] == 64


def test_config5_reports_the_gradient_exchange():
    """BASELINE.json configs[4]: one training step per rank with the gradient exchange of both networks; the dry run trains two small CPU networks
    over gloo through the real launcher, the hook-driven GradientReducer and the reporting code."""
    for world in (1, 2):
        r = _run_bench("--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "config5", "--dry-run", "--hyps", "64")
        assert r.returncode == 0, r.stderr
        d = _json_line(r.stdout)
        ts = d["train_step"]
        assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["ranks_joined"] == world and d["config"]["backend"] == "gloo"
        assert d["config"]["frames_per_step_all_ranks"] == world and ts["step_ms"] > 0 and ts["grad_bytes"] > 0
        if world == 1:
            assert ts["collective_ms"] == 0.0 and ts["collectives_per_step"] == 0
        else:
            assert ts["collective_ms"] > 0 and ts["collectives_per_step"] >= 2


def test_rank_count_mismatch_is_refused():
    r = _run_bench("--gpus", "4", "--steps", "1", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_short_runs_time_every_k2_launch():
    b = _bench()
    assert b.event_stride_for(20, -1) == 1 and b.event_stride_for(64, -1) == 1   # the driver's --steps 20: 20 samples, not 3
    assert b.event_stride_for(200, -1) == 3 and b.event_stride_for(200, 8) == 8 and b.event_stride_for(20, 0) == 0


_FALLBACK_RANK = r'''
import importlib.util, os, sys, types
rank = int(sys.argv[1])
os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
import torch
torch.cuda.device_count = lambda: 2          # pretend two GPUs are visible: the RCCL init then fails for real (there is none here)
torch.cuda.set_device = lambda i: None
spec = importlib.util.spec_from_file_location("bench_module", sys.argv[3])
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
args = types.SimpleNamespace(gpus=2, dry_run=False, workload="default")
r, lr, world, backend, dist = b.init_distributed(args)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("RESULT", backend, world, t.item(), "|", b.BACKEND_NOTE)
dist.destroy_process_group()
'''


def test_timing_collectives_fall_back_to_gloo_when_rccl_cannot_start(tmp_path):
    """The default workload has no data-path collective; if RCCL cannot be initialised on a node the barrier / max of the timing
    scalars go over gloo and the JSON says so (config3, which gathers results with the backend, refuses instead)."""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_FALLBACK_RANK)
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ)
    env.pop("DSAC_BENCH_BACKEND", None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, os.path.join(ROOT, "bench.py")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT")][0]
        assert line.startswith("RESULT gloo 2 2.0") and "RCCL init failed" in line, line

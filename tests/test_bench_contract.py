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


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r01_bench_default.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
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

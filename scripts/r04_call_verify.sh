cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04fu; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_process_images.py tests/test_gpu_forward.py tests/test_gpu_shard.py tests/test_gpu_drivers.py tests/test_gpu_host_shim.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench2.err | tail -1 > $O/bench_head.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04fu/bench_head.json").read())
pi=d.get("process_image",{})
print("value %.0f ms/step %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), {k:round(v["us_per_image"],1) for k,v in pi.items() if isinstance(v,dict) and "us_per_image" in v})
print(d["host_driver"]["us_per_image"], d["host_driver"]["training"]["us_per_frame"], d["cpu_baseline"]["value"])
PY

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04m2; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench2.err | tail -1 > $O/bench_default_streamed.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04m2/bench_default_streamed.json").read())
print("value", d["value"], "roofline", d["roofline"]["frac"])
for k,v in d.get("process_image",{}).items():
    if isinstance(v,dict): print(k, round(v["us_per_image"],1), v.get("refine_steps_done", v.get("refine_steps_done_min")))
print("host_driver", d.get("host_driver"))
PY
for r in 0 7; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline 2>>$O/bench2.err | tail -1 > $O/config3_emulated_w8_rank$r.json; done
for w in 4 2; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world $w --no-cpu-baseline 2>>$O/bench2.err | tail -1 > $O/config3_emulated_w$w.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04m2/config3_emulated_w*.json")):
    d=json.loads(open(f).read()); e=d["emulation"]
    print(f.split("/")[-1], "one_gpu_ms", round(e["one_gpu_ms"],3), "per_rank_ms", round(e["per_rank_ms"],4), "speedup", round(e["predicted_speedup"],2), "k2", round(e["k2_us_per_launch"],1),
          {k:round(v["us_per_image"],1) for k,v in d.get("process_image",{}).items() if isinstance(v,dict) and "us_per_image" in v})
PY
tail -2 $O/bench2.err

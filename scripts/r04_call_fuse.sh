#!/bin/bash
# K7 at the end of K6's wave, score tail in one launch: the tests that pin them, then the latency legs of the bench line.
# (Historical: the one-launch score tail behind DSAC_FUSE_SCORE_TAIL was measured with this script -- profiles/r04_score_tail_fusion_ab.txt -- and then
# removed from the library; the variable has no effect any more.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04fu; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_process_images.py tests/test_gpu_forward.py tests/test_gpu_shard.py tests/test_gpu_drivers.py tests/test_gpu_host_shim.py tests/test_gpu_pipeline.py tests/test_gpu_timed_configs.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
for fz in 1 0 1 0; do
  DSAC_FUSE_SCORE_TAIL=$fz timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver 2>$O/bench.err | tail -1 > $O/bench_fuse$fz.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04fu/bench_fuse$fz.json").read())
pi=d.get("process_image",{})
print("fuse $fz: value %.0f ms/step %.4f frac %.3f | single_frame %s | %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.get("single_frame",{}).items() if k in ("us_per_frame","value","k2_us")},
      {k:round(v["us_per_image"],1) for k,v in pi.items() if isinstance(v,dict) and "us_per_image" in v}))
PY
done

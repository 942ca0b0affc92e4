#!/bin/bash
# two alternating tail streams in "pi_defer_tail" = 2: tests, then the loop of single images and the batch loops again
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tl; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_process_images.py tests/test_gpu_shard.py tests/test_gpu_drivers.py tests/test_gpu_host_shim.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
( cd $O
for rep in 1 2; do for d in 0 1 2; do echo -n "eval batch 1 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 1 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 1: //;s/(.*//'; done; done
for d in 0 1 2; do echo -n "eval batch 1 defer $d, 40x40: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 40 -mh 40 -batch 1 -passes 40 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 1: //;s/(.*//'; done
for d in 0 1 2; do echo -n "eval batch 2 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 2 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 2: //;s/(.*//'; done
for d in 0 1 2; do echo -n "eval batch 4 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 4 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 4: //;s/(.*//'; done
for d in 0 1 2; do echo -n "eval batch 16 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 16 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 16: //;s/(.*//'; done
) | tee $O/tails_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04tl/bench.json").read())
print("value %.0f frac %.3f" % (d["value"], d["roofline"]["frac"]), {k:round(v["us_per_image"],1) for k,v in d["process_image"].items() if isinstance(v,dict) and "us_per_image" in v})
PY
timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_w8.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04tl/config3_w8.json").read()); e=d["emulation"]
print("config3 value %.0f one_gpu_ms %.3f per_rank_ms %.4f speedup %.2f" % (d["value"], e["one_gpu_ms"], e["per_rank_ms"], e["predicted_speedup"]))
PY

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tr; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_host_shim.py -m gpu -q 2>&1 | tail -4 | tee $O/pytest2.log
( cd $O
for d in 0 1 2 0 1 2; do echo -n "eval batch 16 defer $d warm: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 16 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 16: //;s/(.*//'; done
for d in 0 1 2; do echo -n "eval batch 1 defer $d warm: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 1 -passes 12 -defer $d -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 1: //;s/(.*//'; done
for F in 1 8 16; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 100 -batch $F -gradstats 0 -warmup 300 2>&1 | grep -E "Timing|error"; done
for F in 1 16 32; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 40 -mh 40 -rI 256 -rounds 300 -batch $F -gradstats 0 -warmup 300 2>&1 | grep -E "Timing|error"; done
) | tee $O/drivers_warm.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench3.err | tail -1 > $O/bench_default3.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04tr/bench_default3.json").read())
print("value", d["value"], "frac", d["roofline"]["frac"]); print(json.dumps(d.get("host_driver"))[:1500])
PY

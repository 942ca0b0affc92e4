#!/bin/bash
# "pi_defer_tail" = 2 (score tail under the next batch too): the tests that pin it, then the A/B of modes 0 / 1 / 2 on configs[3] (one GPU and the
# emulated rank of 8) and on the C++ evaluation driver.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04m2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_process_images.py tests/test_gpu_shard.py tests/test_gpu_host_shim.py tests/test_gpu_drivers.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
for m in 2 1 2 1; do
  echo "== config3, defer mode $m, emulated rank 0 of 8"
  DSAC_BENCH_DEFER_MODE=$m timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_mode${m}_$RANDOM.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04m2/config3_mode*.json")):
    try:
        d=json.loads(open(f).read())
    except Exception as e:
        print(f, "unparsable", e); continue
    em=d.get("emulated_rank") or d.get("config",{}).get("emulated_rank") or {}
    pi=d.get("process_image") or {}
    print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "emulated per_rank_ms", em.get("per_rank_ms"), "k2", em.get("k2_us_per_launch"),
          {k:round(v["us_per_image"],1) for k,v in pi.items() if isinstance(v,dict) and "us_per_image" in v})
PY
echo "== C++ host driver"; ( cd $O && for i in 1 2; do $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 16 -passes 6 2>&1 | grep -E "Timing" ; done ) | tee $O/host_driver.txt
tail -3 $O/bench.err

#!/bin/bash
# tail streams at the highest stream priority (DSAC_TAIL_PRIO=1) against the default priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tp; mkdir -p $O
for rep in 1 2; do for p in 0 1; do
  DSAC_TAIL_PRIO=$p timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/c_$p.json
  python - <<PY
import json
c=json.loads(open("gpurun_out/r04tp/c_$p.json").read()); e=c["emulation"]
print("DSAC_TAIL_PRIO=$p: config3 value %.0f one_gpu %.3f ms per_rank %.4f ms speedup %.2f K2 %.1f" % (c["value"], e["one_gpu_ms"], e["per_rank_ms"], e["predicted_speedup"], e["k2_us_per_launch"]))
PY
  ( cd $O; for b in 1 16; do echo -n "  eval batch $b: "; DSAC_TAIL_PRIO=$p $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 12 -warmup 300 2>&1 | grep Timing | sed "s/.*batches of $b: //;s/(.*//"; done )
done; done | tee $O/tail_prio.txt

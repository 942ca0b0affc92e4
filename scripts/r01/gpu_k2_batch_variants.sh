#!/bin/bash
# K2 variants inside the default (8 frames per step) bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/frame  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3/d["config"]["frames_per_step"], d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
for v in -1 4 6 7 9 5 0; do for o in 0 1; do
  r=$(DSAC_K2_VARIANT=$v DSAC_K2_ORDER=$o timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --event-stride 1 2>/dev/null | tail -1 | python -c "$fmt")
  echo "variant $v order $o: $r"
done; done | tee gpurun_out/k2_batch_variants.txt

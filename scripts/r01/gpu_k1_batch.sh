#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/frame  %.3f Mhyp/s  frac %.3f" % (d["ms_per_step"]*1e3/d["config"]["frames_per_step"], d["value"]/1e6, d["roofline"]["frac"]))'
for w in 1 4 8 16; do
  r=$(DSAC_K1_WPB=$w timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt")
  echo "batch default, K1 wpb $w: $r"
done | tee gpurun_out/k1_batch.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
for m in err both; do for v in -1 0 4 7 5; do for o in 0 1; do
  r=$(DSAC_K2_VARIANT=$v DSAC_K2_ORDER=$o timeout 300 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --event-stride 1 --k2-mode $m 2>/dev/null | tail -1 | python -c "$fmt")
  echo "N=4096 $m variant $v order $o: $r"
done; done; done | tee gpurun_out/k2_big.txt

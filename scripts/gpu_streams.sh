#!/bin/bash
# default (gated) bench with 2, 3, 4 contexts per GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
for s in 2 3 4; do for o in gated frames; do
  r=$(timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --streams $s --overlap $o 2>/dev/null | tail -1 | python -c "$fmt")
  echo "streams $s overlap $o: $r"
done; done | tee gpurun_out/streams.txt

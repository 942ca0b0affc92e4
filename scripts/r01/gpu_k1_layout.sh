#!/bin/bash
# Default bench (2 contexts, K2 gated) for K1 workgroup sizes / wave priorities
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"]))'
for w in ${K1_WPBS:-1 4 8}; do for p in ${K1_PRIOS:-0 3}; do
  r=$(DSAC_K1_WPB=$w DSAC_K1_PRIO=$p timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt")
  echo "K1 wpb $w prio $p: $r"
done; done | tee gpurun_out/k1_layout.txt
for w in ${K1_WPBS:-1 4 8}; do
  r=$(DSAC_K1_WPB=$w timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --streams 1 --overlap frames --event-stride 1 2>/dev/null | tail -1 | python -c "$fmt")
  echo "single stream, K1 wpb $w: $r"
done | tee -a gpurun_out/k1_layout.txt

#!/bin/bash
# frames batched per step: throughput and K2 roofline fraction vs batch size and overlap mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/frame  %.3f Mhyp/s  frac %.3f  K2 %.1f us/launch" % (d["ms_per_step"]*1e3/d["config"]["frames_per_step"], d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
for b in ${BATCHES:-1 2 4 8 16}; do for m in "1 frames" "2 gated" "2 frames"; do
  set -- $m
  r=$(timeout 300 python bench.py --steps $((400 / b + 20)) --warmup 10 --no-cpu-baseline --frames-per-step $b --streams $1 --overlap $2 2>&1 | tail -1 | python -c "$fmt" 2>&1 | tail -1)
  echo "frames/step $b, streams $1 overlap $2: $r"
done; done | tee gpurun_out/batch.txt

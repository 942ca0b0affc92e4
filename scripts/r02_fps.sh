#!/bin/bash
# frames per launch of the default step: 8 (round 1's choice) against 16 / 32 after the K1 work
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us (n=%d)" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_timed"]))'
{
for fps in 8 16 32 8 16 32; do
  r=$(timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-single-frame --frames-per-step $fps 2>/dev/null | tail -1 | python -c "$fmt")
  echo "frames/step $fps: $r"
done
} | tee $O/frames_per_step.txt

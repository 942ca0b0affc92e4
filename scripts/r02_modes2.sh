#!/bin/bash
# bench modes again after the K1 work (K1 60 -> 35 us per 2048 hypotheses): does overlapping it pay now?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us (n=%d)" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_timed"]))'
{
for mode in "--overlap none" "--overlap pipeline" "--streams 2 --overlap gated"; do
 for fps in 8 1; do
  r=$(timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame --frames-per-step $fps $mode 2>/dev/null | tail -1 | python -c "$fmt")
  echo "frames/step $fps $mode: $r"
 done
done
} | tee $O/bench_modes2.txt

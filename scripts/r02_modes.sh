#!/bin/bash
# Round 2: full GPU suite, then the bench modes (which one becomes the default?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -6 | tee $O/pytest_gpu_full.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us (n=%d)" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_timed"]))'
{
for mode in "--overlap none" "--overlap pipeline" "--streams 2 --overlap gated" "--streams 2 --overlap frames"; do
 for fps in 8 16; do
  r=$(timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame --frames-per-step $fps $mode 2>/dev/null | tail -1 | python -c "$fmt")
  echo "frames/step $fps $mode: $r"
 done
done
for mode in "--overlap none" "--overlap pipeline" "--streams 2 --overlap gated"; do
  r=$(timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-single-frame --frames-per-step 1 $mode 2>/dev/null | tail -1 | python -c "$fmt")
  echo "frames/step 1 $mode: $r"
done
for w in 1 4 8; do
  r=$(DSAC_K1_WPB=$w timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame --overlap pipeline 2>/dev/null | tail -1 | python -c "$fmt")
  echo "pipeline, K1 waves per workgroup $w: $r"
done
} | tee $O/bench_modes.txt

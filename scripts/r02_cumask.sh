#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
{
r=$(timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt"); echo "no overlap: $r"
for cus in 0 16 24 32 48 64 96; do
  r=$(DSAC_K1_CUS=$cus timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame --overlap pipeline 2>/dev/null | tail -1 | python -c "$fmt")
  echo "pipeline, K1 confined to $cus CUs: $r"
done
for cus in 8 16 32; do
  r=$(DSAC_K1_CUS=$cus timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-single-frame --overlap pipeline --frames-per-step 1 2>/dev/null | tail -1 | python -c "$fmt")
  echo "single frame per step, pipeline, K1 confined to $cus CUs: $r"
done
} | tee $O/k1_cumask.txt

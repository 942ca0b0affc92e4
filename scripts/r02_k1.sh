#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
{
for cfg in "DSAC_K1_WPB=4" "DSAC_K1_WPB=1" "DSAC_K1_WPB=8" "DSAC_K1_WPB=4 DSAC_K1_MINW=2" "DSAC_K1_WPB=1 DSAC_K1_MINW=2"; do
  r=$(env $cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt")
  echo "default mode, $cfg: $r"
  env $cfg python scripts/k1_bench.py 2>/dev/null | tail -3
done
} | tee $O/k1_variants.txt
timeout 600 python -m pytest tests/test_gpu_timed_configs.py -m gpu -q -x --timeout 600 -k "two_contexts or unknown" 2>&1 | tail -3

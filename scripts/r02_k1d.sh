#!/bin/bash
# K1: one lane per attempt (k1_rl 1) against one lane per root (4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
DSAC_K1_RL=1 timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_edge.py tests/test_gpu_timed_configs.py -m gpu -q -x --timeout 600 -k "not kernel_form" 2>&1 | tail -3
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us | single frame %.1f us/frame %.3f Mhyp/s | 40x40: %.1f us/frame, 32 frames/step %.2f us/frame" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["single_frame"]["us_per_frame"], d["single_frame"]["value"]/1e6, d["reference_size"]["frames_per_step_1"]["us_per_frame"], d["reference_size"]["frames_per_step_32"]["us_per_frame"]))'
{
for rl in 1 4; do
  r=$(DSAC_K1_RL=$rl timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt")
  echo "default mode, K1 lanes per attempt $rl: $r"
  DSAC_K1_RL=$rl python scripts/k1_bench.py 2>/dev/null | tail -5
done
} | tee $O/k1_rl.txt

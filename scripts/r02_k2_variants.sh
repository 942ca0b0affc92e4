#!/bin/bash
# Round 2: K2 kernel forms (old matrix-core forms 4..9 vs the hypothesis-pair packed forms 20..27) in the three timed shapes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_timed_configs.py -m gpu -q -x --timeout 600 -k "kernel_form" 2>&1 | tail -5 | tee $O/pytest_variants.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
{
for v in -1 20 21 22 23 24 25 26 27; do
  r=$(DSAC_K2_VARIANT=$v timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt")
  echo "batch8 both variant $v: $r"
done
for m in both err; do for v in -1 0 20 21 22 25; do
  r=$(DSAC_K2_VARIANT=$v timeout 300 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>/dev/null | tail -1 | python -c "$fmt")
  echo "N=4096 $m variant $v: $r"
done; done
for m in both err; do for v in -1 0 20 21 23 24; do
  r=$(DSAC_K2_VARIANT=$v timeout 300 python bench.py --steps 200 --warmup 20 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>/dev/null | tail -1 | python -c "$fmt")
  echo "N=256 $m variant $v: $r"
done; done
} | tee $O/k2_variants.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
{
for fl in 0 2; do
  r=$(DSAC_K2_FLAGS=$fl timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt"); echo "batch8 (K1+K2+K3 step), k2_flags=$fl: $r"
  r=$(DSAC_K2_FLAGS=$fl DSAC_K2_VARIANT=21 timeout 300 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --k2-mode both 2>/dev/null | tail -1 | python -c "$fmt"); echo "N=4096 K2 only back to back, k2_flags=$fl: $r"
done
} | tee $O/k2_store_ceiling.txt

#!/bin/bash
# cost of OpenCV's alignment (Horn / Jacobi) in K1 now that an attempt is one lane: scripts/k1_bench.py with the triad and with k1_horn = 1, and the default step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
{
for rep in 1 2; do
  echo "== K1 triad"; timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
  echo "== K1 horn"; DSAC_K1_HORN=1 timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
done
for m in 0 1 0 1; do
  DSAC_K1_HORN=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-driver --no-single-frame 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('horn $m: %.1f us/step %.3f Mhyp/s K2 %.1f us, per-image / kernel-only rate %.3f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['avg_launch_us'], d['rates']['per_image_hyp_s']/d['rates']['kernel_only_k2_hyp_s']))"
done
} > $O/r05_k1_horn_cost.txt 2>&1
cat $O/r05_k1_horn_cost.txt

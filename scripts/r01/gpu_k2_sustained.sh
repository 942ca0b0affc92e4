#!/bin/bash
# K2 'both' variants under SUSTAINED back-to-back launches (clocks settle lower than in isolated runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for v in ${K2_VARIANTS:--1 0 4 6 5}; do for o in ${K2_ORDERS:-0 1}; do
  r=$(DSAC_K2_VARIANT=$v DSAC_K2_ORDER=$o timeout 300 python bench.py --steps 400 --warmup 50 --kernel-only --no-cpu-baseline --streams 1 --event-stride 1 --k2-mode both 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/step  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))")
  echo "variant $v order $o both sustained: $r"
done; done | tee gpurun_out/k2_sustained.txt
# the same variants inside the default bench (2 contexts, K2 gated)
for v in ${K2_VARIANTS:--1 0 4 6 5}; do
  r=$(DSAC_K2_VARIANT=$v DSAC_K2_ORDER=1 timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/step  %.3f Mhyp/s  frac %.3f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['frac']))")
  echo "default bench, variant $v order 1: $r"
done | tee -a gpurun_out/k2_sustained.txt

#!/bin/bash
# K1 with one lane per attempt at a register budget for TWO waves per SIMD (292 B of scratch per lane) against the default (one wave per SIMD, no scratch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04k1; mkdir -p $O
for rep in 1 2; do for m in 1 2; do echo "== DSAC_K1_MINW=$m"; DSAC_K1_MINW=$m timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"; done; done | tee $O/k1_minw.txt
for m in 1 2 1 2; do DSAC_K1_MINW=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-driver --no-single-frame 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MINW=$m: %.1f us/step %.3f Mhyp/s K2 %.1f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['avg_launch_us']))"; done | tee -a $O/k1_minw.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for fps in 16 8 12 4 1; do timeout 300 python bench.py --frames-per-step $fps --steps $((2400/fps)) --warmup 10 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('auto policy, frames/step $fps: %.3f M hyp/s  %.1f us/step  K2 %.1f us frac %.3f' % (d['value']/1e6, d['ms_per_step']*1e3, r['avg_launch_us'], r['frac']))"; done; done

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fps in ${FPS:-2 4 6}; do for vf in 45:32 58:32 45:32 58:32 42:0; do set -- ${vf%%:*} ${vf##*:}; DSAC_K2_VARIANT=$1 DSAC_K2_FLAGS=$2 timeout 300 python bench.py --frames-per-step $fps --steps 150 --warmup 15 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('frames/step $fps variant $1:$2: %.1f us/step  K2 %.1f us frac %.3f' % (d['ms_per_step']*1e3, r['avg_launch_us'], r['frac']))"; done; done

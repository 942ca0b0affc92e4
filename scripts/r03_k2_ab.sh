#!/bin/bash
# K2 A/B in the bench itself (round 3): the lab (scripts/micro/k2_lab, plain HIP process) and bench.py (torch's process) rank the forms differently
# usage: r03_k2_ab.sh [-1 for one frame per step] variant[:flags] ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03
EXTRA=""; if [ "$1" = "-1" ]; then EXTRA="--frames-per-step 1 --steps 300 --warmup 30"; shift; else EXTRA="--steps 40 --warmup 5"; fi
b() { v=${1%%:*}; f=0; [[ "$1" == *:* ]] && f=${1##*:}; DSAC_K2_VARIANT=$v DSAC_K2_FLAGS=$f timeout 300 python bench.py $EXTRA --no-cpu-baseline --no-single-frame 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"bench variant $1: %.1f us/step  K2 %.1f us frac %.3f store-only %.1f\" % (d[\"ms_per_step\"]*1e3, r[\"avg_launch_us\"], r[\"frac\"], r[\"store_schedule_only_us\"] or 0))"; }
for v in "$@"; do b $v; done

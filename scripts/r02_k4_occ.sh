#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
for v in 2 1 321 322 221 421 312 311 411; do echo "== k4_variant $v"; DSAC_K4_VARIANT=$v timeout 300 python scripts/k4_bench.py 2>&1 | grep "K4 N" | grep d_err; done | tee $O/k4_occupancy.txt

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python scripts/k2_sweep.py 2>&1 | tail -40 | tee gpurun_out/k2_sweep.log
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -5

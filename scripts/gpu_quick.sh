#!/bin/bash
# quick GPU iteration: tests + optional sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python scripts/k2_sweep.py 2>&1 | tail -40 | tee gpurun_out/k2_sweep.log
for s in 1 2 4; do timeout 300 python bench.py --steps 200 --warmup 20 --streams $s --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330; done | tee gpurun_out/bench_streams.log

#!/bin/bash
# Round 2, first GPU pass: the GPU test suite (incl. the new timed-configuration tests), the bench with the driver's flags and the default
# flags, K4 stage timing.  Outputs -> gpurun_out/r02/
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -40 | tee $O/pytest_gpu.log
echo "== bench, driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-600
echo "== bench, default"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-600
echo "== K4"; timeout 300 python scripts/k4_bench.py 2>&1 | grep "K4 N" | tee $O/k4_bench.log

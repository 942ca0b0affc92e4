#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace. Outputs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== nproc: $(nproc)"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== diag"; timeout 300 python scripts/diag_sample.py 2>&1 | tail -30 | tee gpurun_out/diag.log
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== bench kernel-only N=4096"
timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_k2_4096.log
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o k -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > /tmp/prof.log 2>&1; tail -3 /tmp/prof.log )
find /tmp/prof -type f | head -10
for f in $(find /tmp/prof -name '*kernel_stats*.csv' | head -1); do cp "$f" gpurun_out/kernel_stats.csv; head -12 "$f"; done

#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python scripts/k4_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/k4_bench_fused.log; cat gpurun_out/r04/k4_bench_fused.log
DSAC_K4_VARIANT=1999 python scripts/k4_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/k4_bench_legacy.log; cat gpurun_out/r04/k4_bench_legacy.log
python scripts/k4_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/k4_bench_fused2.log; cat gpurun_out/r04/k4_bench_fused2.log
python scripts/diag_k4_soft.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/diag_k4_soft.txt; head -9 gpurun_out/r04/diag_k4_soft.txt
DSAC_MARGINS_FILE=gpurun_out/r04/parity_margins.txt python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu.log 2>&1; tail -12 gpurun_out/r04/pytest_gpu.log

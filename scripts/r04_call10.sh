#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python scripts/diag_k4_soft.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/diag_k4_soft.txt; head -12 gpurun_out/r04/diag_k4_soft.txt
python -m pytest tests/test_gpu_backward.py tests/test_gpu_backward_big.py tests/test_gpu_pipeline.py tests/test_gpu_reference_golden.py tests/test_gpu_e2e.py tests/test_gpu_drivers.py -q -x 2>&1 | tail -12
python scripts/k4_bench.py > gpurun_out/r04/k4_bench_fused.log 2>&1; cat gpurun_out/r04/k4_bench_fused.log | tail -20
DSAC_K4_VARIANT=1999 python scripts/k4_bench.py > gpurun_out/r04/k4_bench_legacy.log 2>&1; cat gpurun_out/r04/k4_bench_legacy.log | tail -20

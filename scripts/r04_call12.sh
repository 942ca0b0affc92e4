#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_backward.py tests/test_gpu_backward_big.py tests/test_gpu_pipeline.py tests/test_gpu_reference_golden.py tests/test_gpu_e2e.py tests/test_gpu_edge.py -q -x 2>&1 | tail -4
for i in 1 2; do
python scripts/k4_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/k4_bench_fused_$i.log; cat gpurun_out/r04/k4_bench_fused_$i.log
DSAC_K4_VARIANT=1999 python scripts/k4_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/k4_bench_legacy_$i.log; cat gpurun_out/r04/k4_bench_legacy_$i.log
done

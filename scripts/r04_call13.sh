#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_backward_batch.py tests/test_gpu_backward.py tests/test_gpu_refine.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_dsac_variant.py tests/test_gpu_timed_configs.py -q -x 2>&1 | tail -25
python scripts/train_geometry_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04/train_geometry.txt; cat gpurun_out/r04/train_geometry.txt

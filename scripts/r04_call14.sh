#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_backward_batch.py -q -x -s 2>&1 | tail -30
python -m pytest tests/test_gpu_backward.py tests/test_gpu_refine.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_dsac_variant.py tests/test_gpu_timed_configs.py -q -x 2>&1 | tail -8

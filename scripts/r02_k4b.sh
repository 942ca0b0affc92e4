#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_pipeline.py tests/test_gpu_reference_golden.py -m gpu -q --timeout 600 2>&1 | grep -E "FAILED|passed|failed|Error|rel " | head -30 | tee $O/pytest_k4.log
for v in -1 2 3 4 5; do echo "== k4_variant $v (chunks per wave: 2 -> 4, 3 -> 5, 4 -> 6, 5 -> 3, -1 auto)"; DSAC_K4_VARIANT=$v timeout 300 python scripts/k4_bench.py 2>&1 | grep "K4 N"; done | tee $O/k4_chunks.txt

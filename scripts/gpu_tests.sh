#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -${1:-30} | tee gpurun_out/pytest_gpu.log

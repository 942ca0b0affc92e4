#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 0 --no-cpu-baseline 2>gpurun_out/r04/e.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps(j['emulation'], indent=1))"
tail -3 gpurun_out/r04/e.err
python scripts/diag_k4_soft.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/diag_k4_soft.txt; cat gpurun_out/r04/diag_k4_soft.txt
DSAC_MARGINS_FILE=gpurun_out/r04/parity_margins.txt python -m pytest tests -m gpu -q --deselect tests/test_gpu_backward_big.py > gpurun_out/r04/pytest_gpu.log 2>&1; tail -15 gpurun_out/r04/pytest_gpu.log

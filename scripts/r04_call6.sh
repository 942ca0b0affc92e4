#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_shard.py -x -q 2>&1 | tail -2
python scripts/r04_rank_step_lab.py 8 60 4 > gpurun_out/r04/rank_step_lab_8.txt 2>&1; cat gpurun_out/r04/rank_step_lab_8.txt
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 0 --no-cpu-baseline > gpurun_out/r04/config3_emulated_rank0.json 2> gpurun_out/r04/config3_emulated_rank0.err
tail -2 gpurun_out/r04/config3_emulated_rank0.err

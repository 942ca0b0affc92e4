#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_shard.py -x -q > gpurun_out/r04/shard_tests.log 2>&1; tail -3 gpurun_out/r04/shard_tests.log
for r in 0 7; do
  python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline > gpurun_out/r04/config3_emulated_rank$r.json 2> gpurun_out/r04/config3_emulated_rank$r.err
  tail -3 gpurun_out/r04/config3_emulated_rank$r.err
done
python scripts/r04_k2_err_ab.py 6 > gpurun_out/r04/k2_err_ab.txt 2>&1; cat gpurun_out/r04/k2_err_ab.txt

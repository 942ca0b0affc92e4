#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_process_images.py -x -q > gpurun_out/r04/pi_tests.log 2>&1; tail -3 gpurun_out/r04/pi_tests.log
for r in 0 7; do
  python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline > gpurun_out/r04/config3_emulated_rank$r.json 2> gpurun_out/r04/config3_emulated_rank$r.err
  tail -3 gpurun_out/r04/config3_emulated_rank$r.err; cat gpurun_out/r04/config3_emulated_rank$r.json
done
DSAC_BENCH_NO_DEFER=1 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline > gpurun_out/r04/config3_emulated_nodefer.json 2>&1
cat gpurun_out/r04/config3_emulated_nodefer.json
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 4 --no-cpu-baseline > gpurun_out/r04/config3_emulated_w4.json 2>&1
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 2 --no-cpu-baseline > gpurun_out/r04/config3_emulated_w2.json 2>&1
python scripts/r04_k2_err_ab.py 3 > gpurun_out/r04/k2_err_ab.txt 2>&1; cat gpurun_out/r04/k2_err_ab.txt

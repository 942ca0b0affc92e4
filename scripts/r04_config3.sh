#!/bin/bash
# round 4: ShardRunner parity tests + configs[3] on one GPU with the one-GPU emulation of rank r of 8
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_shard.py tests/test_gpu_process_images.py -x -q -s > gpurun_out/r04/shard_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04/shard_tests.log
tail -15 gpurun_out/r04/shard_tests.log
for r in 0 7; do
  python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline > gpurun_out/r04/config3_emulated_rank$r.json 2> gpurun_out/r04/config3_emulated_rank$r.err
  tail -3 gpurun_out/r04/config3_emulated_rank$r.err; cat gpurun_out/r04/config3_emulated_rank$r.json
done
DSAC_BENCH_NO_DEFER=1 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline > gpurun_out/r04/config3_emulated_nodefer.json 2>&1
cat gpurun_out/r04/config3_emulated_nodefer.json
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 4 --no-cpu-baseline > gpurun_out/r04/config3_emulated_w4.json 2>&1
python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 2 --no-cpu-baseline > gpurun_out/r04/config3_emulated_w2.json 2>&1

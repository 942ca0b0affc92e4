#!/bin/bash
# round 3: the bench line in its workloads (default with soft_only / process_image batch, config3 = whole processImage, config5 = training step)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O
echo "== default, driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-400
echo "== config3"; timeout 600 python bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline 2>$O/bench_c3.err | tail -1 | tee $O/bench_config3.json | cut -c1-600
echo "== config5, 1 GPU"; timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 2>$O/bench_c5.err | tail -1 | tee $O/bench_config5.json | cut -c1-1500
echo "== config5, 2 ranks on this one GPU over gloo (launcher / reducer check, not a scaling number)"; DSAC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload config5 --steps 5 --warmup 2 2>$O/bench_c5g.err | tail -1 | tee $O/bench_config5_2ranks_gloo.json | cut -c1-1500
tail -3 $O/*.err

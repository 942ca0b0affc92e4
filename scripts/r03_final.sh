#!/bin/bash
# Round 3 closing run at HEAD: GPU tests, smoke, the bench line in its workloads, rocprofv3 kernel trace + PMC of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03f; mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench, driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-300
echo "== bench, default flags"; timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | tail -1 | tee $O/bench_default.json | cut -c1-200
echo "== config3"; timeout 600 python bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/bench.err | tail -1 | tee $O/bench_config3.json | cut -c1-200
echo "== config5"; timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 2>>$O/bench.err | tail -1 | tee $O/bench_config5.json | cut -c1-200
echo "== config5 x2 ranks over gloo on this one GPU"; DSAC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload config5 --steps 5 --warmup 2 2>>$O/bench.err | tail -1 | tee $O/bench_config5_2ranks_gloo.json | cut -c1-200
echo "== K2 only: configs[2] N=4096, both / err / soft"; for m in both err soft; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>>$O/bench.err | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-160; done
echo "== K4"; timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N" | tee $O/k4_bench.log
GRAFT_OUT=$O scripts/r03_round_profiles.sh > $O/profiles.log 2>&1; tail -12 $O/profiles.log | cut -c1-300
cp gpurun_out/r03p/* $O/ 2>/dev/null
tail -3 $O/bench.err

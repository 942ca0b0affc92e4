#!/bin/bash
# Round 4 closing run at HEAD: GPU tests with the parity margins kept, smoke, the bench line in its workloads, the C++ host driver, the one-GPU
# emulation of configs[3] at 8 ranks, the K4 stage (fused / round-3 staging), the training geometry on frame batches, rocprofv3 kernel trace of the
# driver's bench command and of the K4 stage, PMC traffic of K2 (writes profiles/k2_traffic.json's successor itself).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04f; mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
echo "== pytest -m gpu -q -s (margins -> $O/parity_margins.txt)"
DSAC_MARGINS_FILE=$REPO/$O/parity_margins.txt timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_full.log 2>&1; tail -3 $O/pytest_gpu_full.log | tee $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench, driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-300
echo "== bench, default flags"; timeout 600 python bench.py --no-cpu-baseline --no-host-driver 2>>$O/bench.err | tail -1 | tee $O/bench_default.json | cut -c1-200
echo "== config3 + emulation of rank 0 / 7 of 8, rank 0 of 4 and 2"
for r in 0 7; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline 2>>$O/bench.err | tail -1 | tee $O/config3_emulated_w8_rank$r.json | cut -c1-200; done
for w in 4 2; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world $w --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w$w.json; done
DSAC_BENCH_NO_DEFER=1 timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w8_tail_in_order.json
echo "== config5"; timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 2>>$O/bench.err | tail -1 | tee $O/bench_config5.json | cut -c1-200
echo "== K2 only: configs[2] N=4096, both / err / soft"; for m in both err soft; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>>$O/bench.err | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-160; done
echo "== C++ host driver"; ( cd $O && for b in 16 0; do $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 12 -warmup 300 2>&1 | grep -E "Timing|Avg|Median" ; done ) | tee $O/host_driver.txt
echo "== K4 stage, fused (default) and the round-3 staging, alternating"; for i in 1 2; do timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N" | sed "s/^/fused   /"; DSAC_K4_VARIANT=1999 timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N" | sed "s/^/staging /"; done | tee $O/k4_stage.txt
echo "== training geometry on frame batches"; timeout 900 python scripts/train_geometry_bench.py 2>&1 | grep "device-resident" | tee $O/train_geometry.txt
echo "== rank step lab"; timeout 600 python scripts/r04_rank_step_lab.py 8 60 4 2>&1 | grep -v amdgpu | tee $O/rank_step_lab_8.txt
export TMPDIR=/tmp; cd /tmp
echo "== rocprofv3 kernel trace of the driver's bench command"
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv; head -8 /tmp/kt/k_kernel_stats.csv | cut -c1-220
grep "^{" /tmp/kt.log | tail -1 > $REPO/$O/bench_driver_flags_under_rocprof.json
echo "== rocprofv3 kernel trace of the K4 stage"
rm -rf /tmp/k4; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o k -- python $REPO/scripts/k4_bench.py > /tmp/k4.log 2>&1
cp /tmp/k4/k_kernel_stats.csv $REPO/$O/k4_stage_kernel_stats.csv; head -8 /tmp/k4/k_kernel_stats.csv | cut -c1-220
echo "== rocprofv3 kernel trace of the training geometry (frame batches)"
rm -rf /tmp/tg; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg -o k -- python $REPO/scripts/train_geometry_bench.py > /tmp/tg.log 2>&1
cp /tmp/tg/k_kernel_stats.csv $REPO/$O/train_geometry_kernel_stats.csv; head -12 /tmp/tg/k_kernel_stats.csv | cut -c1-200
cd $REPO
echo "== PMC traffic of K2"; bash scripts/r04_k2_pmc.sh > $O/k2_pmc.log 2>&1; cp gpurun_out/r04/k2_pmc.txt $O/ 2>/dev/null; cp gpurun_out/r04/k2_traffic.json $O/ 2>/dev/null; tail -2 $O/k2_pmc.log | cut -c1-400
tail -3 $O/bench.err

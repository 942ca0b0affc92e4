#!/bin/bash
# Round 5 closing run at HEAD: GPU tests with the parity margins kept, smoke, the driver's bench command plain and under rocprofv3, the `strong` object from the
# one-GPU emulation, configs[3] the old way (cross-check), config5, K2 alone at configs[2], the C++ programs (built-in score and through the seam), the precise
# A/B, the K4 stage, the training geometry, PMC traffic of K2 (k2_traffic.json's successor).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r05final; mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
echo "== pytest -m gpu -q -s (margins -> $O/parity_margins.txt)"
DSAC_MARGINS_FILE=$REPO/$O/parity_margins.txt timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; tail -3 $O/pytest_gpu_full.log | tee $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench, driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-300
echo "== bench, default flags"; timeout 600 python bench.py --no-cpu-baseline --no-host-driver 2>>$O/bench.err | tail -1 | tee $O/bench_default.json | cut -c1-200
echo "== the strong object: one GPU emulating rank 0 / 7 of 8 (and the score tail in order for the A/B)"
for r in 0 7; do timeout 600 python bench.py --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline --no-host-driver --no-single-frame 2>>$O/bench.err | tail -1 > $O/bench_em8_rank$r.json; python -c "
import json; d=json.loads(open('$O/bench_em8_rank$r.json').read()); s=d['strong']; print('rank $r of 8: one_gpu %.3f ms per_rank %.3f ms speedup %.2f' % (s['one_gpu_ms'], s['per_rank_ms'], s['speedup']))"; done
DSAC_BENCH_NO_DEFER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver --no-single-frame 2>>$O/bench.err | tail -1 > $O/bench_score_tail_in_order.json
echo "== config3 (cross-check of the leg) and config5"
timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w8_rank0.json
timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 2>>$O/bench.err | tail -1 | tee $O/bench_config5.json | cut -c1-200
echo "== K2 only: configs[2] N=4096, both / err / soft"; for m in both err soft; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>>$O/bench.err | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-160; done
echo "== C++ host programs"; ( cd $O && for b in 16 0; do $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 12 -warmup 300 2>&1 | grep -E "Timing|Avg|Median" ; done; for s in 0 1; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 60 -batch 16 -gradstats 0 -warmup 300 -seam $s 2>&1 | grep Timing; done ) | tee $O/host_driver.txt
echo "== K2 default / precise / two-piece records A/B"; timeout 600 python scripts/r05_k2_precise_ab.py 2>&1 | grep -v amdgpu | tee $O/k2_precise_ab.txt | head -8
echo "== K4 stage"; timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N" | tee $O/k4_stage.txt
echo "== training geometry on frame batches"; timeout 900 python scripts/train_geometry_bench.py 2>&1 | grep "device-resident" | tee $O/train_geometry.txt
echo "== DSAC variant on frame batches"; timeout 600 python scripts/dsac_variant_bench.py 2>&1 | grep "DSAC variant" | tee $O/dsac_variant.txt
export TMPDIR=/tmp; cd /tmp
echo "== rocprofv3 kernel trace of the driver's bench command"
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv; head -8 /tmp/kt/k_kernel_stats.csv | cut -c1-220
grep "^{" /tmp/kt.log | tail -1 > $REPO/$O/bench_driver_flags_under_rocprof.json
cd $REPO
echo "== PMC traffic of K2"; bash scripts/r04_k2_pmc.sh > $O/k2_pmc.log 2>&1; cp gpurun_out/r04/k2_pmc.txt $O/ 2>/dev/null; cp gpurun_out/r04/k2_traffic.json $O/ 2>/dev/null; tail -2 $O/k2_pmc.log | cut -c1-400
tail -3 $O/bench.err

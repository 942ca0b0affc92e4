#!/bin/bash
# Round 6 closing run at HEAD: GPU tests with the parity margins kept, smoke, the driver's bench command plain and under rocprofv3, the `strong` object from the
# one-GPU emulation, K2 forms A/B, K6 walk, K4 stage, training geometry, DSAC variant, the C++ programs, PMC traffic of the default (exact) K2.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r06final; mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
echo "== pytest -m gpu -q -s (margins -> $O/parity_margins.txt)"
DSAC_MARGINS_FILE=$REPO/$O/parity_margins.txt timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; tail -3 $O/pytest_gpu_full.log | tee $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench, driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 | tee $O/bench_driver_flags.json | cut -c1-300
echo "== bench, driver flags, on the fp32 form (for the record)"; timeout 900 python bench.py --steps 20 --warmup 5 --k2-form fast --no-cpu-baseline --no-host-driver --no-k2-forms 2>>$O/bench.err | tail -1 | tee $O/bench_driver_flags_fast_form.json | cut -c1-200
echo "== the strong object: one GPU emulating rank 0 / 7 of 8"
for r in 0 7; do timeout 600 python bench.py --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline --no-host-driver --no-single-frame --no-k2-forms 2>>$O/bench.err | tail -1 > $O/bench_em8_rank$r.json; python -c "
import json; d=json.loads(open('$O/bench_em8_rank$r.json').read()); s=d['strong']; print('rank $r of 8: one_gpu %.3f ms per_rank %.3f ms speedup %.2f' % (s['one_gpu_ms'], s['per_rank_ms'], s['speedup']))"; done
echo "== K2 only: configs[2] N=4096, both / err / soft"; for m in both err soft; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>>$O/bench.err | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-160; done
echo "== C++ host programs"; ( cd $O && for b in 16 0; do $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 12 -warmup 300 2>&1 | grep -E "Timing|Avg|Median" ; done; $REPO/dsac_amd/host/test_ransac_softam -synth 16 -mw 640 -mh 480 -batch 0 -refstream 1 -passes 3 2>&1 | grep -E "Timing|Avg|reference random"; for s in 0 1; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 60 -batch 16 -gradstats 0 -warmup 300 -seam $s 2>&1 | grep Timing; done ) | tee $O/host_driver.txt
echo "== K2 forms A/B"; DSAC_AB_MODES=fast,precise,exact DSAC_AB_ROUNDS=3 DSAC_AB_ROUNDS2=1 timeout 600 python scripts/r06_k2_exact_ab.py 2>&1 | grep -v amdgpu | tee $O/k2_exact_ab.txt | head -12
echo "== K6 walk"; DSAC_K6_WAVES=1,4,0,-100 timeout 600 python scripts/micro/k6_walk_bench.py 2>&1 | grep -v amdgpu | tee $O/k6_walk.txt | head -40
echo "== K6 scan: kernel trace"; MODES=0 timeout 400 bash scripts/micro/k6_walk_trace.sh | tee $O/k6_walk_trace.txt
echo "== K4 stage"; timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N" | tee $O/k4_stage.txt
echo "== training geometry on frame batches"; timeout 900 python scripts/train_geometry_bench.py 2>&1 | grep "device-resident" | tee $O/train_geometry.txt
echo "== DSAC variant on frame batches"; timeout 600 python scripts/dsac_variant_bench.py 2>&1 | grep "DSAC variant" | tee $O/dsac_variant.txt
export TMPDIR=/tmp; cd /tmp
echo "== rocprofv3 kernel trace of the driver's bench command"
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver --no-k2-forms > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv; head -8 /tmp/kt/k_kernel_stats.csv | cut -c1-220
grep "^{" /tmp/kt.log | tail -1 > $REPO/$O/bench_driver_flags_under_rocprof.json
cd $REPO
echo "== PMC traffic of K2 (the default, exact form)"; DSAC_PMC_EXTRA="--no-k2-forms" bash scripts/r04_k2_pmc.sh > $O/k2_pmc.log 2>&1; cp gpurun_out/r04/k2_pmc.txt $O/ 2>/dev/null; cp gpurun_out/r04/k2_traffic.json $O/ 2>/dev/null; tail -2 $O/k2_pmc.log | cut -c1-400
tail -3 $O/bench.err

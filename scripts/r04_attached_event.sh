#!/bin/bash
# the tail's start event attached to K2's dispatch instead of a record behind it: tests, rank step, batch loop, single-image loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04ae; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_process_images.py tests/test_gpu_shard.py tests/test_gpu_drivers.py tests/test_gpu_host_shim.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.log
( cd $O
for rep in 1 2; do for b in 1 16; do echo -n "eval batch $b defer 2: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 12 -defer 2 -warmup 300 2>&1 | grep Timing | sed "s/.*batches of $b: //;s/(.*//"; done; done
echo -n "eval batch 1 defer 2, 40x40: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 40 -mh 40 -batch 1 -passes 40 -defer 2 -warmup 300 2>&1 | grep Timing | sed 's/.*batches of 1: //;s/(.*//'
) | tee $O/attached.txt
for rep in 1 2; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/err.log | tail -1 | python -c "
import json,sys; c=json.loads(sys.stdin.read()); e=c['emulation']; print('config3 value %.0f one_gpu %.3f ms per_rank %.4f ms speedup %.2f' % (c['value'], e['one_gpu_ms'], e['per_rank_ms'], e['predicted_speedup']))"; done | tee -a $O/attached.txt

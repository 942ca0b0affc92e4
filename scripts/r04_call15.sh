#!/bin/bash
mkdir -p gpurun_out/r04
for v in "" 1 "" 1; do
DSAC_BENCH_EM_SHARE_ERR=$v python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=j['emulation']; print('share_err=%s' % '$v', '1gpu K2 %.1f' % j['roofline']['avg_launch_us'], 'per_rank %.4f k2 %.1f eff %.3f' % (e['per_rank_ms'], e['k2_us_per_launch'], e['predicted_efficiency']))"
done
cd gpurun_out/r04 && ../../dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 1 -passes 6 2>&1 | grep -E "Timing"

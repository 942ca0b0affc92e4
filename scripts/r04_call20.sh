#!/bin/bash
for v in "" 1 "" 1; do
DSAC_BENCH_EM_SHARE_ERR=$v python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=j['emulation']; print('share=%s' % '$v', '1gpu K2 %.1f ms %.3f' % (j['roofline']['avg_launch_us'], j['ms_per_step']), 'per_rank %.4f k2 %.1f eff %.3f' % (e['per_rank_ms'], e['k2_us_per_launch'], e['predicted_efficiency']))"
done

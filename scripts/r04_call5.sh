#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python scripts/r04_rank_step_lab.py 8 60 4 > gpurun_out/r04/rank_step_lab_8.txt 2>&1; cat gpurun_out/r04/rank_step_lab_8.txt
python scripts/r04_rank_step_lab.py 16 40 2 > gpurun_out/r04/rank_step_lab_16.txt 2>&1; cat gpurun_out/r04/rank_step_lab_16.txt
for m in both err; do python bench.py --kernel-only --hyps 4096 --k2-mode $m --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$m', j['roofline']['avg_launch_us'], j['roofline']['frac'], j['roofline'].get('store_schedule_only_us'))"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default', j['value'], j['roofline']['avg_launch_us'], j['roofline']['frac'], j['process_image'])"

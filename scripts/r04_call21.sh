#!/bin/bash
O=gpurun_out/r04f2; mkdir -p $O
for r in 0 7; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w8_rank$r.json; done
for w in 4 2; do timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world $w --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w$w.json; done
DSAC_BENCH_NO_DEFER=1 timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/config3_emulated_w8_tail_in_order.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04f2/config3_emulated_*.json')):
    j=json.loads([l for l in open(f) if l.startswith('{')][-1]); e=j['emulation']
    print(f.split('/')[-1], '1gpu %.3f ms K2 %.1f | per_rank %.4f ms k2 %.1f host %.3f speedup %.2f eff %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_us'], e['per_rank_ms'], e['k2_us_per_launch'], e['host_enqueue_ms_per_step'], e['predicted_speedup'], e['predicted_efficiency']))
PY

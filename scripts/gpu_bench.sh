#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
export TMPDIR=/tmp
for s in 1 2 3 4; do timeout 300 python bench.py --steps 300 --warmup 30 --streams $s --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('streams', j['config']['streams_per_gpu'], 'value %.0f hyp/s' % j['value'], 'ms/step %.4f' % j['ms_per_step'], 'k2 us %.1f' % j['roofline']['avg_launch_us'], 'frac %.3f' % j['roofline']['frac'])
"; done | tee gpurun_out/bench_streams.log
( cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o k -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --streams 1 --no-cpu-baseline > /tmp/prof.log 2>&1 )
python - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/prof/k_kernel_stats.csv")))
for r in rows[:8]:
    print("%-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cp /tmp/prof/k_kernel_stats.csv gpurun_out/kernel_stats_streams1.csv

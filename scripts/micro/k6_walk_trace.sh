#!/bin/bash
# Kernel trace of the walk bench's 128-problem case (k_refine_walk / k_refine_lm / k_refine_permute durations). Output -> gpurun_out/r06_k6_walk_trace.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-0 -100}; do
  rm -rf /tmp/k6t
  DSAC_K6_CASE="128 problems" DSAC_K6_WAVES=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k6t -o k -- python $REPO/scripts/micro/k6_walk_bench.py > /tmp/k6t.log 2>&1
  echo "== DSAC_K6_WAVES=$mode (-100: the walk's fp32 filter off)"
  grep "K6 " /tmp/k6t.log
  f=$(find /tmp/k6t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:6]: print('  %-28s calls %5s  avg %10.1f us  total %10.1f us  %5s %%' % (r['Name'].split('(')[0].replace('void ','')[:28], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3, r['Percentage']))
"
done 2>&1 | tee $REPO/gpurun_out/r06_k6_walk_trace.txt

#!/bin/bash
# Per-kernel breakdown of one image per call (exact K2, and the fp32 form for comparison). Output -> gpurun_out/r06_one_image_trace.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $REPO/gpurun_out; cd /tmp; export TMPDIR=/tmp
for off in "" 1; do
  rm -rf /tmp/oit
  DSAC_K2_EXACT_AUTO_OFF=$off timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oit -o k -- python $REPO/scripts/micro/one_image_trace.py > /tmp/oit.log 2>&1
  echo "== K2 form: ${off:+fp32 }${off:-exact}"; grep "one image" /tmp/oit.log
  f=$(find /tmp/oit -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 -c "
import csv
tot=0
for r in list(csv.DictReader(open('$f')))[:9]:
    print('  %-34s calls %5s  avg %8.1f us' % (r['Name'].split('(')[0].replace('void ','')[:34], r['Calls'], float(r['AverageNs'])/1e3)); 
"
done 2>&1 | tee $REPO/gpurun_out/r06_one_image_trace.txt

#!/bin/bash
# Where the C++ training program (train_ransac_softam -batch 16: 205 us per frame) spends more than the Python geometry bench (163 us per frame):
# rocprofv3 kernel statistics of both, same shape (16 frames x 256 hypotheses x 640x480 per round / step).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=$REPO/gpurun_out/r05t; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/tg_cpp /tmp/tg_py /tmp/tg_run; mkdir -p /tmp/tg_run
(cd /tmp/tg_run && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg_cpp -o t -- $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 60 -batch 16 -gradstats 0 -warmup 300 > $O/train_cpp.log 2>&1)
grep Timing $O/train_cpp.log
DSAC_TGB_ONLY=480x640x16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg_py -o t -- python $REPO/scripts/train_geometry_bench.py > $O/train_py.log 2>&1
grep "device-resident" $O/train_py.log
for t in cpp py; do f=$(find /tmp/tg_$t -name "*kernel_stats.csv" | head -1); cp $f $O/train_${t}_kernel_stats.csv; echo "== $t"; python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s  total %10.1f us  avg %8.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done

#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --event-stride 0 --prewarm-ms 0"
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- $K2F > /tmp/pmc_$c.log 2>&1
  python - /tmp/pmc_$c/p_counter_collection.csv $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_reproject" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
done

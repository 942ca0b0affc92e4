#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for cus in 0 16; do
  rm -rf /tmp/ktp$cus; DSAC_K1_CUS=$cus timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp$cus -o k -- python $REPO/bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-single-frame --overlap pipeline > /tmp/ktp$cus.log 2>&1
  echo "== pipeline, k1_cus=$cus"; head -5 /tmp/ktp$cus/k_kernel_stats.csv | cut -c1-60,200-330
  python - /tmp/ktp$cus/k_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k2 = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_reproject_hp" in r["Kernel_Name"]][-40:]
k1 = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_sample" in r["Kernel_Name"]][-40:]
print("K2 dur mean %.1f us; gap between consecutive K2 launches mean %.1f us" % (sum(e - s for s, e in k2) / len(k2) / 1e3, sum(k2[i + 1][0] - k2[i][1] for i in range(len(k2) - 1)) / (len(k2) - 1) / 1e3))
print("K1 dur mean %.1f us" % (sum(e - s for s, e in k1) / len(k1) / 1e3))
# overlap of each K1 with K2 launches
ov = 0
for s, e in k1:
    for s2, e2 in k2:
        ov += max(0, min(e, e2) - max(s, s2))
print("K1 time overlapped by K2: %.1f us per launch" % (ov / len(k1) / 1e3))
PY
done 2>&1 | tee $REPO/$O/pipeline_trace.txt

#!/bin/bash
# kernel timeline of one round of the C++ training program (-batch 16): where the time between kernels goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tr; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o k -- $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 12 -batch ${1:-16} -gradstats 0 > /tmp/tr.log 2>&1
python - <<'PY' | tee $REPO/gpurun_out/r04tr/train_timeline.txt
import csv, glob
f = glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# rounds start at k_gather_rows (first of the three)
starts = [i for i, n in enumerate(names) if "k_gather_rows" in n and (i == 0 or "k_gather_rows" not in names[i - 1])]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    busy += e - s; prev_end = max(prev_end, e)
print("round period %.1f us, kernels %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3))
PY
grep Timing /tmp/tr.log

#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $REPO/bench.py --steps 30 --warmup 10 --no-cpu-baseline --event-stride 0 "$@" > /tmp/tl.log 2>&1
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open("/tmp/tl/t_kernel_trace.csv")) if "dk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:len(rows)//2+24]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dk::", "")[:22]
    print("%-22s q%-3s start %8.1f  end %8.1f  dur %6.1f" % (n, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY

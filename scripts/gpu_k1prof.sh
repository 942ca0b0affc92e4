#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 4 --no-cpu-baseline --streams 1 --overlap frames --event-stride 0"
rm -rf /tmp/p1; timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d /tmp/p1 -o p -- $CMD > /tmp/p1.log 2>&1
python - <<'PY'
import csv, collections
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open("/tmp/p1/p_kernel_trace.csv"))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("/tmp/p1/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0][:40]
    if "dk::" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); agg[k]["dur_ns"].append(dur[r["Dispatch_Id"]]); agg[k]["VGPR"].append(float(r["VGPR_Count"])); agg[k]["AGPR"].append(float(r["Accum_VGPR_Count"]))
for k, d in agg.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(d.items())))
PY

#!/bin/bash
# SQ counters of the walk bench's 128-problem case (k_refine_walk / k_refine_lm). Output -> gpurun_out/r06_k6_walk_pmc.txt
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
CMD="python $REPO/scripts/micro/k6_walk_bench.py"
rm -rf /tmp/p1; DSAC_K6_CASE="128 problems" DSAC_K6_WAVES=0 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -o p -- $CMD > /tmp/p1.log 2>&1
grep "K6 " /tmp/p1.log
python - <<'PY' | tee $REPO/gpurun_out/r06_k6_walk_pmc.txt
import csv, collections, glob
kt = glob.glob("/tmp/p1/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob("/tmp/p1/**/*counter_collection.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0][:40]
    if "dk::" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); agg[k]["dur_ns"].append(dur[r["Dispatch_Id"]]); agg[k]["VGPR"].append(float(r["VGPR_Count"]))
for k, d in agg.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(d.items())))
PY

#!/bin/bash
# PMC passes on the K4 main-pass forms (N = 256, 640x480)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
pmc() { # tag counters -- cmd
  tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag/p_counter_collection.csv /tmp/pmc_$tag/p_kernel_trace.csv $tag <<'PY'
import csv, sys, collections
cc, kt, tag = sys.argv[1:4]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if "k_score_backward" in k or "k_reproject" in k:
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k.split("(")[0][:60]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for k, d in agg.items():
    print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
{
for v in 1 2; do for m in d_err; do
  export DSAC_K4_VARIANT=$v
  pmc k4v${v}_${m}_sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -- python $REPO/scripts/k4_one.py 256 $m 6
  pmc k4v${v}_${m}_b GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS -- python $REPO/scripts/k4_one.py 256 $m 6
done; done
} 2>&1 | tee $REPO/$O/k4_pmc.txt | cut -c1-400

#!/bin/bash
# K4 main pass under the PMC counters (round 3): where do the cycles go?
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r03p; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $REPO/$O/sq_counters_available.txt
pmc() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag/p_counter_collection.csv /tmp/pmc_$tag/p_kernel_trace.csv $tag <<'PY'
import csv, sys, collections
cc, kt, tag = sys.argv[1:4]
try:
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        if "k_score_backward" in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k.split("(")[0][:60]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
    for k, d in agg.items():
        print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
except Exception as e:
    print(tag, "failed:", e)
PY
}
K4="python $REPO/scripts/k4_one.py 256 d_err 6"
{
pmc k4_a SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 -- $K4
pmc k4_b GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD -- $K4
pmc k4_c SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT -- $K4
pmc k4_d SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_WAVE32 -- $K4
} 2>&1 | tee $REPO/$O/k4_pmc.txt | cut -c1-600
cat $REPO/$O/sq_counters_available.txt | cut -c1-3000

#!/bin/bash
# K4 main pass (k_score_backward_mfma, N = 256 x 640x480, explicit d_err) under the SQ counters: where its wave cycles go -- issuing (ACTIVE_INST_ANY), stalled at
# issue (WAIT_INST_ANY: pipe / dependency), parked on s_waitcnt or a barrier (WAIT_ANY) --, how much of the launch the SQ is busy at all (ramp + tail), and the
# VMEM issue cycles.  Separate --pmc passes with --kernel-trace only (gpurun refuses PMC together with other trace domains).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r05k4; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
pmc() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag/p_counter_collection.csv $tag <<'PY'
import csv, sys, collections
cc, tag = sys.argv[1:3]
try:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("k_score_backward", "k_support_scatter")):
            agg[k.split("(")[0][:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print("%s | %-52s | n=%d | " % (tag, k, len(next(iter(d.values())))) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
except Exception as e:
    print(tag, "failed:", e, open("/tmp/pmc_%s.log" % tag).read()[-400:])
PY
}
K4="python $REPO/scripts/k4_one.py 256 d_err 8"
{
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -- $K4
pmc sq2 SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -- $K4
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT -- $K4
} 2>&1 | tee $REPO/$O/k4_sq_pmc.txt | cut -c1-400
cd "$REPO"; timeout 200 python scripts/k4_bench.py 2>&1 | grep "K4 N" | tee $O/k4_bench.txt

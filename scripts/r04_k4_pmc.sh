#!/bin/bash
# K4 stage under the PMC counters, fused (round 4) against the round-3 staging: HBM traffic (WRITE_SIZE + 2 x FETCH_SIZE, KiB) and VALU instructions
# of the main pass, separate --pmc passes with --kernel-trace only.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
pmc() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag/p_counter_collection.csv $tag <<'PY'
import csv, sys, collections
cc, tag = sys.argv[1:3]
try:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("k_score_backward", "k_support_scatter", "k_grad_reduce", "k_backward_prep")):
            agg[k.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print("%s | %-48s | n=%d | " % (tag, k, len(next(iter(d.values())))) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
except Exception as e:
    print(tag, "failed:", e)
PY
}
K4="python $REPO/scripts/k4_one.py 256 d_err 6"
{
pmc fused_write WRITE_SIZE -- $K4
pmc fused_fetch FETCH_SIZE -- $K4
pmc fused_valu SQ_INSTS_VALU SQ_WAVES -- $K4
DSAC_K4_VARIANT=1999 pmc staging_write WRITE_SIZE -- $K4
DSAC_K4_VARIANT=1999 pmc staging_fetch FETCH_SIZE -- $K4
DSAC_K4_VARIANT=1999 pmc staging_valu SQ_INSTS_VALU SQ_WAVES -- $K4
} 2>&1 | tee $REPO/$O/k4_pmc.txt | cut -c1-300

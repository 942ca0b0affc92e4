#!/bin/bash
# PMC passes (separate WRITE_SIZE / FETCH_SIZE runs, kernel trace only) of the K2 launch of the default step (16 frames x 256 hypotheses)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
pmc() { # tag counters -- cmd
  tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag; timeout 600 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag/p_counter_collection.csv /tmp/pmc_$tag/p_kernel_trace.csv $tag <<'PY'
import csv, sys, collections
cc, kt, tag = sys.argv[1:4]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if "k_reproject" in k:
        agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k.split("(")[0][:70]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for k, d in agg.items():
    print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --event-stride 0 --prewarm-ms 0"
{
pmc k2_batch16_write WRITE_SIZE -- $K2F
pmc k2_batch16_fetch FETCH_SIZE -- $K2F
} 2>&1 | tee $REPO/$O/pmc16.txt | cut -c1-300

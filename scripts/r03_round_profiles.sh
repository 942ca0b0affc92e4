#!/bin/bash
# Round 3 artefacts: rocprofv3 kernel-trace stats of the driver's bench command, PMC traffic of the new default K2 form (16-frame batch), K4 PMC.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "== rocprofv3 kernel trace of the driver's bench command"
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv; head -8 /tmp/kt/k_kernel_stats.csv | cut -c1-220
grep "^{" /tmp/kt.log | tail -1 > $REPO/$O/bench_driver_flags_under_rocprof.json
python - /tmp/kt/k_kernel_trace.csv > $REPO/$O/bench_driver_flags_k2_launches.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byk = collections.defaultdict(list)
for r in rows:
    if "k_reproject" in r["Kernel_Name"]:
        byk[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in byk.items():
    if len(d) < 100:
        continue
    # the batch kernel: prewarm + 5 warm-up + 20 timed + 6 store-only + 1 + soft-only ... ; the 20 timed launches follow the warm-up
    print("%s: %d launches in the trace, mean of all %.1f us" % (k[:90], len(d), sum(d) / len(d) / 1e3))
PY
cat $REPO/$O/bench_driver_flags_k2_launches.txt
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
    if "k_reproject" in k or "k_score_backward" in k:
        agg[k.split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k.split("(")[0][:80]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for k, d in agg.items():
    print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --event-stride 0 --prewarm-ms 0"
K21="python $REPO/bench.py --steps 30 --warmup 5 --frames-per-step 1 --no-cpu-baseline --no-single-frame --event-stride 0 --prewarm-ms 0"
{
pmc k2_batch16_write WRITE_SIZE -- $K2F
pmc k2_batch16_fetch FETCH_SIZE -- $K2F
pmc k2_batch16_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -- $K2F
pmc k2_batch16_clk GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -- $K2F
pmc k2_single_write WRITE_SIZE -- $K21
pmc k2_single_fetch FETCH_SIZE -- $K21
} 2>&1 | tee $REPO/$O/pmc_summary.txt | cut -c1-400

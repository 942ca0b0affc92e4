#!/bin/bash
# Round artefacts: bench line, rocprofv3 kernel-trace stats, PMC traffic of K2/K4.  Outputs -> gpurun_out/r01/
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; O=gpurun_out/r01; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
echo "== bench 2 contexts, frames round-robin (K2 launches may overlap)"; timeout 600 python bench.py --frames-per-step 1 --streams 2 --overlap frames --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_frames2.json | cut -c1-200
echo "== bench in-context pipeline"; timeout 600 python bench.py --frames-per-step 1 --streams 1 --overlap pipeline --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_pipeline.json | cut -c1-200
echo "== bench (default)"; timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-400
echo "== bench single frame per step, 2 contexts gated"; timeout 600 python bench.py --frames-per-step 1 --streams 2 --overlap gated --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_single_gated2.json | cut -c1-200
echo "== bench 8 frames per step, 2 contexts gated"; timeout 600 python bench.py --frames-per-step 8 --streams 2 --overlap gated --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_batch8_gated2.json | cut -c1-200
echo "== bench single frame per step, no overlap"; timeout 600 python bench.py --frames-per-step 1 --streams 1 --overlap frames --event-stride 1 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_streams1.json | cut -c1-200
echo "== bench K2 only, N=4096 (configs[2])"; for m in err both; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --event-stride 1 --k2-mode $m 2>/dev/null | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-120; done
echo "== bench K2 only, N=256"; for m in err both; do timeout 600 python bench.py --steps 200 --warmup 20 --kernel-only --no-cpu-baseline --streams 1 --event-stride 1 --k2-mode $m 2>/dev/null | tail -1 | tee $O/bench_k2only_256_$m.json | cut -c1-120; done
cd /tmp
echo "== rocprofv3 kernel trace of the default bench"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_kernel_stats.csv; head -7 /tmp/kt/k_kernel_stats.csv | cut -c1-160
tail -1 /tmp/kt.log | cut -c1-300 > $REPO/$O/bench_under_rocprof.json
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
        agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k.split("(")[0][:70]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for k, d in agg.items():
    print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
K2="python $REPO/bench.py --steps 12 --warmup 4 --kernel-only --no-cpu-baseline --streams 1 --event-stride 0 --k2-mode both"
K2E="python $REPO/bench.py --steps 12 --warmup 4 --kernel-only --no-cpu-baseline --streams 1 --event-stride 0 --k2-mode err"
K2B="python $REPO/bench.py --steps 6 --warmup 2 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --event-stride 0 --k2-mode both"
{
pmc k2_256_both_write WRITE_SIZE -- $K2
pmc k2_256_both_fetch FETCH_SIZE -- $K2
pmc k2_256_err_write WRITE_SIZE -- $K2E
pmc k2_256_err_fetch FETCH_SIZE -- $K2E
pmc k2_4096_both_write WRITE_SIZE -- $K2B
pmc k2_4096_both_fetch FETCH_SIZE -- $K2B
pmc k2_256_both_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -- $K2
pmc k2_256_both_clk GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- $K2
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --event-stride 0"
pmc k2_batch8_write WRITE_SIZE -- $K2F
pmc k2_batch8_fetch FETCH_SIZE -- $K2F
pmc k4_write WRITE_SIZE -- python $REPO/scripts/k4_bench.py
pmc k4_fetch FETCH_SIZE -- python $REPO/scripts/k4_bench.py
pmc k4_sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python $REPO/scripts/k4_bench.py
} 2>&1 | tee $REPO/$O/pmc_summary.txt | cut -c1-260
echo "== K4 kernel trace"
rm -rf /tmp/kt4; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o k -- python $REPO/scripts/k4_bench.py > $REPO/$O/k4_bench.log 2>&1
cp /tmp/kt4/k_kernel_stats.csv $REPO/$O/k4_kernel_stats.csv; grep "K4 N" $REPO/$O/k4_bench.log
cd $REPO; ./scripts/micro/store_pattern 2>/dev/null | sort -k11 -n -r | head -12 > $O/store_pattern_top.txt

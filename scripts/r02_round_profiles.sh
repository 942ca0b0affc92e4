#!/bin/bash
# Round 2 artefacts: bench lines (driver flags, default, modes), rocprofv3 kernel-trace stats of the driver command, PMC traffic of K2 / K4,
# kernel stats of the training-path kernels.  Outputs -> gpurun_out/r02p/
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
echo "== bench, driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $O/bench_driver_flags.json | cut -c1-300
echo "== bench (default)"; timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-300
echo "== bench K2 only, N=4096 (configs[2])"; for m in err both; do timeout 600 python bench.py --steps 30 --warmup 5 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>/dev/null | tail -1 | tee $O/bench_k2only_4096_$m.json | cut -c1-120; done
echo "== bench K2 only, N=256"; for m in err both; do timeout 600 python bench.py --steps 200 --warmup 20 --kernel-only --no-cpu-baseline --streams 1 --k2-mode $m 2>/dev/null | tail -1 | tee $O/bench_k2only_256_$m.json | cut -c1-120; done
echo "== bench configs[3] on one GPU"; timeout 600 python bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_config3_1gpu.json | cut -c1-300
echo "== bench --gpus 2 on this one GPU (gloo rendezvous: launcher check)"; DSAC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_gpus2_gloo_oneGPU.json | cut -c1-300
cd /tmp
echo "== rocprofv3 kernel trace of the driver's bench command"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv; head -6 /tmp/kt/k_kernel_stats.csv | cut -c1-200
tail -1 /tmp/kt.log > $REPO/$O/bench_driver_flags_under_rocprof.json
# per-launch durations of K2 inside the timed region (last 20 launches of the batch kernel)
python - /tmp/kt/k_kernel_trace.csv > $REPO/$O/bench_driver_flags_k2_launches.txt <<'PY'
import csv, sys
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_reproject_hp<64, true, true, false, 4>" in r["Kernel_Name"]]
t = d[-120:-100] if len(d) > 140 else d[-20:]
print("k_reproject_hp<64,ERR,SOFT,4> launches in trace: %d ; the 20 timed launches of the batch phase: mean %.1f us min %.1f max %.1f" % (len(d), sum(t) / len(t) / 1e3, min(t) / 1e3, max(t) / 1e3))
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
        agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k.split("(")[0][:70]]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for k, d in agg.items():
    print("%s | %s | n=%d | " % (tag, k, len(d["_dur_ns"])) + " | ".join("%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
K2="python $REPO/bench.py --steps 12 --warmup 4 --kernel-only --no-cpu-baseline --streams 1 --event-stride 0 --k2-mode both --prewarm-ms 0"
K2B="python $REPO/bench.py --steps 6 --warmup 2 --hyps 4096 --kernel-only --no-cpu-baseline --streams 1 --event-stride 0 --k2-mode both --prewarm-ms 0"
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --event-stride 0 --prewarm-ms 0"
{
pmc k2_256_both_write WRITE_SIZE -- $K2
pmc k2_256_both_fetch FETCH_SIZE -- $K2
pmc k2_4096_both_write WRITE_SIZE -- $K2B
pmc k2_4096_both_fetch FETCH_SIZE -- $K2B
pmc k2_batch8_write WRITE_SIZE -- $K2F
pmc k2_batch8_fetch FETCH_SIZE -- $K2F
pmc k2_batch8_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -- $K2F
pmc k2_batch8_clk GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -- $K2F
pmc k4_write WRITE_SIZE -- python $REPO/scripts/k4_one.py 256 d_err 6
pmc k4_fetch FETCH_SIZE -- python $REPO/scripts/k4_one.py 256 d_err 6
pmc k4_sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 -- python $REPO/scripts/k4_one.py 256 d_err 6
pmc k4_clk GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -- python $REPO/scripts/k4_one.py 256 d_err 6
} 2>&1 | tee $REPO/$O/pmc_summary.txt | cut -c1-300
echo "== K4 kernel trace (whole stage)"
rm -rf /tmp/kt4; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o k -- python $REPO/scripts/k4_bench.py > $REPO/$O/k4_bench.log 2>&1
cp /tmp/kt4/k_kernel_stats.csv $REPO/$O/k4_kernel_stats.csv; grep "K4 N" $REPO/$O/k4_bench.log
echo "== training step + DSAC variant kernel trace"
rm -rf /tmp/kt5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -o k -- python $REPO/scripts/train_step_bench.py > $REPO/$O/train_step_bench.log 2>&1
cp /tmp/kt5/k_kernel_stats.csv $REPO/$O/train_step_kernel_stats.csv 2>/dev/null; tail -3 $REPO/$O/train_step_bench.log
rm -rf /tmp/kt6; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt6 -o k -- python $REPO/scripts/dsac_variant_bench.py > $REPO/$O/dsac_variant_bench.log 2>&1
cp /tmp/kt6/k_kernel_stats.csv $REPO/$O/dsac_variant_kernel_stats.csv 2>/dev/null; tail -3 $REPO/$O/dsac_variant_bench.log

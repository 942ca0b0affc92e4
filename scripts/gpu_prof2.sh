#!/bin/bash
# PMC profile of one K2 configuration: env DSAC_K2_VARIANT / DSAC_K2_ORDER / DSAC_K2_FLAGS, K2MODE, HYPS
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 6 --warmup 2 --hyps ${HYPS:-4096} --kernel-only --no-cpu-baseline --k2-mode ${K2MODE:-both}"
run_pmc() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- $CMD > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if "k_reproject" not in k and "k_score_backward" not in k: continue
    agg[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(name, k)
    for c, v in sorted(d.items()):
        print("   %-30s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  else echo "no counter csv for $name"; tail -5 /tmp/pmc_$name.log; fi
}
echo "== config variant=${DSAC_K2_VARIANT:-} order=${DSAC_K2_ORDER:-} flags=${DSAC_K2_FLAGS:-} mode=${K2MODE:-both} hyps=${HYPS:-4096}"
run_pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run_pmc b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU
run_pmc c GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS
run_pmc w WRITE_SIZE
run_pmc f FETCH_SIZE

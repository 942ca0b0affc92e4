#!/bin/bash
# PMC / trace profiling of K2 on the GPU box. Outputs -> gpurun_out/prof_*
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE '\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+' | sort -u > $REPO/gpurun_out/counters.txt
wc -l $REPO/gpurun_out/counters.txt
CMD="python $REPO/bench.py --steps 6 --warmup 2 --hyps 4096 --kernel-only --no-cpu-baseline --k2-mode ${K2MODE:-err}"
run_pmc() { # name, counters...
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
    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(name, k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  else echo "no counter csv for $name"; tail -5 /tmp/pmc_$name.log; fi
}
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run_pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum

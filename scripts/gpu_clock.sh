#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 6 --warmup 2 --hyps ${HYPS:-4096} --kernel-only --no-cpu-baseline --k2-mode ${K2MODE:-err}"
rm -rf /tmp/pmc_clk
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_clk -o p -- $CMD > /tmp/pmc_clk.log 2>&1
ls /tmp/pmc_clk
head -2 /tmp/pmc_clk/p_counter_collection.csv | cut -c1-600
head -2 /tmp/pmc_clk/p_kernel_trace.csv | cut -c1-600
python - <<'PY'
import csv, collections
kt = {}
for r in csv.DictReader(open("/tmp/pmc_clk/p_kernel_trace.csv")):
    kt[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for r in csv.DictReader(open("/tmp/pmc_clk/p_counter_collection.csv")):
    if "k_reproject" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        name, dur = kt[r["Dispatch_Id"]]
        v = float(r["Counter_Value"])
        print("dispatch %s dur %.1f us  GRBM_GUI_ACTIVE %.4g  -> clock if /8: %.3f GHz, if /1: %.3f GHz" % (r["Dispatch_Id"], dur / 1e3, v, v / 8 / dur, v / dur))
PY
rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk" | head
rocm-smi --showpower --showmaxpower 2>/dev/null | grep -iE "power" | head -4

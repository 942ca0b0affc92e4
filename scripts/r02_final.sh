#!/bin/bash
# Round-2 closing run: the whole GPU suite, smoke(), the bench with the driver's flags + its rocprofv3 kernel trace.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02f; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_flags.json
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-frame > /tmp/kt.log 2>&1
cp /tmp/kt/k_kernel_stats.csv $REPO/$O/bench_driver_flags_kernel_stats.csv
python - /tmp/kt/k_kernel_trace.csv $REPO/$O/bench_driver_flags.json <<'PY' | tee $REPO/$O/bench_vs_rocprof.txt
import csv, sys, json
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_reproject_hp<64, true, true, false, 4>" in r["Kernel_Name"]]
t = d[-27:-7]  # the timed region's 20 launches; 7 more follow (6 store-only twins + one real launch)
b = json.loads(open(sys.argv[2]).read())
print("rocprofv3: the 20 timed K2 launches: mean %.1f us (min %.1f max %.1f); bench.py HIP events (separate run, same box): %.1f us, frac %.3f, %.3f M hyp/s" % (sum(t) / len(t) / 1e3, min(t) / 1e3, max(t) / 1e3, b["roofline"]["avg_launch_us"], b["roofline"]["frac"], b["value"] / 1e6))
PY

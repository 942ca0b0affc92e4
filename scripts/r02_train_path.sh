#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02p; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/kt7; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt7 -o k -- python $REPO/scripts/train_path_profile.py > $REPO/$O/train_path.log 2>&1
cp /tmp/kt7/k_kernel_stats.csv $REPO/$O/train_path_kernel_stats.csv; grep -v "^[EW]2026" $REPO/$O/train_path.log | tail -5; cut -c1-140 $REPO/$O/train_path_kernel_stats.csv | head -40
cd $REPO; echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_flags_v2.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02p/bench_driver_flags_v2.json").read())
print(d["value"], d["roofline"]["frac"], d["reference_size"], d["cpu_baseline"]["reference_size"])
PY

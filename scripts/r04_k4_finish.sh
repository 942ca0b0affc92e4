#!/bin/bash
# K4's finish kernel (k_support_scatter) after requesting its partial rows ahead of the pose-only fp64 work: tests, stage time, kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04k4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_backward_big.py tests/test_gpu_backward_batch.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2; do timeout 600 python scripts/k4_bench.py 2>&1 | grep "K4 N= 256"; done | tee $O/k4_stage.txt
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/k4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o k -- python $REPO/scripts/k4_bench.py > /tmp/k4.log 2>&1
grep -E "k_support_scatter|k_score_backward_mfma" /tmp/k4/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-400 | tee $REPO/$O/k4_kernels.txt
python - <<'PY'
import csv
for r in csv.DictReader(open("/tmp/k4/k_kernel_stats.csv")):
    if "support_scatter" in r["Name"] or "score_backward_mfma" in r["Name"]:
        print(r["Name"][:50], r["Calls"], "avg %.2f us min %.2f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY

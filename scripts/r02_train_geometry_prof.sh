#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/kt9; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt9 -o k -- python $REPO/scripts/train_geometry_bench.py > $REPO/$O/train_geometry_prof.log 2>&1
cp /tmp/kt9/k_kernel_stats.csv $REPO/$O/train_geometry_kernel_stats.csv; grep "device-resident" $REPO/$O/train_geometry_prof.log; cut -c1-130 $REPO/$O/train_geometry_kernel_stats.csv | head -24

#!/bin/bash
# kernel table of the training path after the K6 work (profiles/r02_train_path_v2_*)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/kt8; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o k -- python $REPO/scripts/train_path_profile.py > $REPO/$O/train_path_v2.log 2>&1
cp /tmp/kt8/k_kernel_stats.csv $REPO/$O/train_path_v2_kernel_stats.csv; grep -v "^[EW]2026" $REPO/$O/train_path_v2.log | tail -4; cut -c1-150 $REPO/$O/train_path_v2_kernel_stats.csv | head -30

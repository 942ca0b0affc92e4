#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("K2 %.1f us  frac %.3f" % (d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))'
for mode in 0 1 2 3 4 5; do
  r=$(DSAC_K2_FLAGS=$((2 + mode*4)) timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt"); echo "store-only, cache policy $mode (0 nt, 1 plain, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0): $r"
done | tee $O/k2_store_policy.txt

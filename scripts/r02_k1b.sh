#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02
for cfg in "DSAC_K1_SHARE=0" "DSAC_K1_SHARE=-4" "DSAC_K1_SHARE=-4 DSAC_K1_MINW=2" "DSAC_K1_SHARE=-8"; do echo "== $cfg"; env $cfg python scripts/k1_bench.py 2>/dev/null | tail -5; done | tee $O/k1_share_minw.txt

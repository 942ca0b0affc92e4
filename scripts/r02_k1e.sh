#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
{
for cfg in "DSAC_K1_HPW=1" "DSAC_K1_HPW=2" "DSAC_K1_RL=4 DSAC_K1_SHARE=4"; do
  echo "== $cfg"; env $cfg python scripts/k1_bench.py 2>/dev/null | tail -5
done
} | tee $O/k1_rl_hpw.txt

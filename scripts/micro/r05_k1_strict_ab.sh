#!/bin/bash
# K1 / K5 built with -ffp-contract=off (the oracle's and OpenCV's arithmetic: no fused multiply-add) against the default build: accepted minimal sets that
# differ from the oracle's in the ill-conditioned narrow-window geometry, and the kernel's time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
NEW=$PWD/dsac_amd/csrc/build/ab/libdsac_hip_k1${NEWLIB:-strict}.so
export DSAC_DIAG_BRIEF=1
{
for L in ${LIBS:-default strict strict_pow}; do
  if [ $L = default ]; then unset DSAC_HIP_LIB; else export DSAC_HIP_LIB=$PWD/dsac_amd/csrc/build/ab/libdsac_hip_k1$L.so; fi
  echo "== $L"; timeout 300 python scripts/micro/r05_k1_accept_diag.py 2>&1 | grep -v "centred\|amdgpu.ids"
done
unset DSAC_HIP_LIB
for rep in 1 2; do
  echo "== K1 default"; timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
  echo "== K1 -ffp-contract=off"; DSAC_HIP_LIB=$NEW timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
done
} > $O/r05_k1_strict_ab.txt 2>&1
cat $O/r05_k1_strict_ab.txt

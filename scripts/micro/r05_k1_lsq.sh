#!/bin/bash
# K1 with strict arithmetic + the closed-form least-squares alignment (the build in the tree): accepted sets / poses against the oracle, K1's time; K5 with
# the same alignment against K5 with OpenCV's Jacobi sweeps (build/ab/libdsac_hip_k5jacobi.so): time and the dPNP margins; then the tests that see K1 / K5
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
export DSAC_DIAG_BRIEF=1
JAC=$PWD/dsac_amd/csrc/build/ab/libdsac_hip_k5jacobi.so
{
timeout 300 python scripts/micro/r05_k1_accept_diag.py 2>&1 | grep -v "centred\|amdgpu.ids\|k1_horn"
for rep in 1 2; do timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"; done
for rep in 1 2; do
  echo "== K5 closed form"; timeout 300 python scripts/micro/k5_bench.py 2>&1 | grep "K5 N"
  echo "== K5 Jacobi"; DSAC_HIP_LIB=$JAC timeout 300 python scripts/micro/k5_bench.py 2>&1 | grep "K5 N"
done
echo "== dPNP margins, closed form"; timeout 600 python -m pytest tests/test_gpu_forward.py -q -m gpu -k dpnp -s 2>&1 | grep "margin\|passed\|failed\|dPNP seed"
echo "== dPNP margins, Jacobi"; DSAC_HIP_LIB=$JAC timeout 600 python -m pytest tests/test_gpu_forward.py -q -m gpu -k dpnp -s 2>&1 | grep "margin\|passed\|failed\|dPNP seed"
} > $O/r05_k1_lsq.txt 2>&1
cat $O/r05_k1_lsq.txt | grep -v "poses on\|differ;"
grep -c "0 of 1024 differ" $O/r05_k1_lsq.txt; grep "total differing" $O/r05_k1_lsq.txt
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_pipeline.py tests/test_gpu_reference_golden.py tests/test_gpu_reference_golden_dsac.py tests/test_gpu_process_images.py tests/test_gpu_random_shapes.py tests/test_gpu_edge.py tests/test_gpu_backward.py tests/test_gpu_dsac_variant.py -x -q -m gpu 2>&1 | tail -8 | tee $O/r05_k1_lsq_tests.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_edge.py tests/test_gpu_timed_configs.py tests/test_gpu_reference_golden.py tests/test_gpu_pipeline.py -m gpu -q -x --timeout 600 -k "not kernel_form" 2>&1 | tail -3
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s | single frame %.1f us/frame | 40x40: %.1f us/frame | process_image %.1f / %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["single_frame"]["us_per_frame"], d["reference_size"]["frames_per_step_1"]["us_per_frame"], d["process_image"]["640x480"]["us_per_image"], d["process_image"]["40x40"]["us_per_image"]))'
{
for wide in 0 2 4; do
  echo "== k1_wide $wide"; DSAC_K1_WIDE=$wide python scripts/k1_bench.py 2>/dev/null | head -3 | tail -2
  DSAC_K1_WIDE=$wide timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt"
done
} | tee $O/k1_wide.txt

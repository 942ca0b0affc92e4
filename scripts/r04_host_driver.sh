#!/bin/bash
# round 4: the C++ host surface's fast path -- tests + timing of the evaluation program on 64 synthetic 640x480 frames
set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_host_shim.py tests/test_gpu_drivers.py -x -q -s > gpurun_out/r04/host_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04/host_tests.log
tail -5 gpurun_out/r04/host_tests.log
cd gpurun_out/r04
for b in 16 0; do
  ../../dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 6 2>&1 | grep -E "Timing|Avg|Median|error" > host_driver_batch$b.txt
  cat host_driver_batch$b.txt
done
../../dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 16 -passes 6 -errimg 0 2>&1 | grep -E "Timing" > host_driver_batch16_noerr.txt; cat host_driver_batch16_noerr.txt

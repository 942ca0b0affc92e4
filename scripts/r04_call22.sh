#!/bin/bash
mkdir -p gpurun_out/r04; cd gpurun_out/r04
for b in 16 16 8 32; do ../../dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch $b -passes 8 2>&1 | grep -E "Timing"; done
cd ../..; python -m pytest tests/test_gpu_drivers.py tests/test_gpu_host_shim.py tests/test_gpu_process_images.py -q 2>&1 | tail -2

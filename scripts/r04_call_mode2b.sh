#!/bin/bash
# C++ evaluation driver, deferred-tail modes 0 / 1 / 2 alternating, in batches of 16 and one image per launch chain
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04m2; mkdir -p $O; cd $O
for rep in 1 2 3; do for d in 0 1 2; do
  echo -n "batch 16 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 16 -passes 12 -defer $d 2>&1 | grep Timing | sed 's/.*batches of 16: //;s/(.*//'
done; done | tee host_driver_defer_ab.txt
for rep in 1 2; do for d in 0 1 2; do
  echo -n "batch 1 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 1 -passes 12 -defer $d 2>&1 | grep Timing | sed 's/.*batches of 1: //;s/(.*//'
done; done | tee -a host_driver_defer_ab.txt
for rep in 1 2; do for d in 0 2; do
  echo -n "batch 4 defer $d: "; $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 4 -passes 12 -defer $d 2>&1 | grep Timing | sed 's/.*batches of 4: //;s/(.*//'
done; done | tee -a host_driver_defer_ab.txt

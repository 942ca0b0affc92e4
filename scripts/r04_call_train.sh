#!/bin/bash
# C++ training program: the device-resident rounds (-batch F) against the per-image loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tr; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_host_shim.py tests/test_gpu_backward_batch.py tests/test_gpu_edge.py -m gpu -q 2>&1 | tail -8 | tee $O/pytest.log
cd $O
for F in 1 8 16; do
  $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 40 -batch $F -gradstats 0 2>&1 | grep -E "Timing|error"
done | tee train_driver.txt
$REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 40 -batch 16 -gradstats 0 -errimg 0 2>&1 | grep -E "Timing|error" | sed 's/^/no error images: /' | tee -a train_driver.txt
for F in 1 16 32; do
  $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 40 -mh 40 -rI 256 -rounds 100 -batch $F -gradstats 0 2>&1 | grep -E "Timing|error"
done | tee -a train_driver.txt
python - <<PY | tee -a train_driver.txt
import subprocess, time
t0 = time.perf_counter()
subprocess.run(["$REPO/dsac_amd/host/train_ransac_softam", "-synth", "8", "-mw", "640", "-mh", "480", "-rI", "256", "-rounds", "10"], capture_output=True)
t1 = time.perf_counter()
subprocess.run(["$REPO/dsac_amd/host/train_ransac_softam", "-synth", "8", "-mw", "640", "-mh", "480", "-rI", "256", "-rounds", "30"], capture_output=True)
t2 = time.perf_counter()
print("per-image loop (Frame::processImage / Frame::backward, host arrays), 640x480: %.1f ms per round (30-round run minus 10-round run)" % (((t2 - t1) - (t1 - t0)) / 20 * 1e3))
PY
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o k -- $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds 40 -batch 16 -gradstats 0 > /tmp/tr.log 2>&1
cp /tmp/tr/k_kernel_stats.csv $REPO/$O/train_driver_f16_kernel_stats.csv; head -25 /tmp/tr/k_kernel_stats.csv | cut -c1-150
grep Timing /tmp/tr.log

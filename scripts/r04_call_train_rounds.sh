cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04tr; mkdir -p $O; cd $O
for R in 40 200 400; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds $R -batch 16 -gradstats 0 2>&1 | grep -E "Timing|error"; done
for R in 100 1000; do $REPO/dsac_amd/host/train_ransac_softam -synth 32 -mw 640 -mh 480 -rI 256 -rounds $R -batch 1 -gradstats 0 2>&1 | grep -E "Timing|error"; done
cd $REPO; python scripts/train_geometry_bench.py 2>&1 | grep "device-resident"

#!/bin/bash
# the literal configs[1] step (ONE 640x480 frame x 256 hypotheses per step) in the bench's overlap modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04sf; mkdir -p $O
for rep in 1 2; do
for m in "none 1" "pipeline 1" "stages 1" "gated 2"; do set -- $m
  timeout 300 python bench.py --frames-per-step 1 --overlap $1 --streams $2 --steps 400 --warmup 40 --no-cpu-baseline --no-host-driver --no-single-frame 2>>$O/err.log | tail -1 > $O/sf_$1.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04sf/sf_$1.json").read())
print("frames/step 1 --overlap $1 --streams $2: %.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
PY
done; done | tee $O/single_frame_modes.txt

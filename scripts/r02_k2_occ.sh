#!/bin/bash
# K2 matrix-core forms under an occupancy cap (unused dynamic LDS): is the slowness of <64,2> an occupancy effect?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f us/step  %.3f Mhyp/s  frac %.3f  K2 %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))'
{
for v in 24 20 21 25 27; do for pad in 0 4 6 9; do
  r=$(DSAC_K2_VARIANT=$v DSAC_K2_FLAGS=$((pad*256)) timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-single-frame 2>/dev/null | tail -1 | python -c "$fmt")
  echo "batch8 both variant $v lds pad ${pad}x8KB: $r"
done; done
} | tee $O/k2_occupancy.txt

#!/bin/bash
# does GPU_MAX_HW_QUEUES (HIP's number of hardware queues per device, default 4) matter for the bench process, which holds more streams than that?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04hq; mkdir -p $O
for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-driver 2>>$O/err.log | tail -1 > $O/b_$q.json
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --workload config3 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/c_$q.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04hq/b_$q.json").read()); c=json.loads(open("gpurun_out/r04hq/c_$q.json").read()); e=c["emulation"]
pi=d["process_image"]
print("GPU_MAX_HW_QUEUES=$q: value %.0f frac %.3f | stream %s / %s | batch16 %s %s %s | config3 one_gpu %.3f ms per_rank %.4f ms speedup %.2f" % (d["value"], d["roofline"]["frac"],
  round(pi["640x480_stream_of_images_refinement_under_the_next_image"]["us_per_image"],1), round(pi["640x480_stream_of_images_score_and_refinement_under_the_next_image"]["us_per_image"],1),
  round(pi["640x480_batch_of_16"]["us_per_image"],1), round(pi["640x480_batch_of_16_refinement_under_the_next_batch"]["us_per_image"],1), round(pi["640x480_batch_of_16_score_and_refinement_under_the_next_batch"]["us_per_image"],1),
  e["one_gpu_ms"], e["per_rank_ms"], e["predicted_speedup"]))
PY
done | tee $O/hwq.txt

#!/bin/bash
mkdir -p gpurun_out/r04; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/tl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/scripts/r04_rank_timeline.py > /tmp/tl.log 2>&1
python $R/scripts/r04_rank_timeline.py /tmp/tl/t_kernel_trace.csv | tee $R/gpurun_out/r04/rank_timeline.txt
tail -3 /tmp/tl.log

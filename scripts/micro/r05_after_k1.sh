#!/bin/bash
# after the K1 / K5 numerics change (strict P3P arithmetic inside dmath.h, closed-form least-squares alignment in K1, K5 and the DSAC variant's replica starts):
# the whole GPU suite, the training-geometry and DSAC-variant benches, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05k1; mkdir -p $O
export DSAC_MARGINS_FILE=$PWD/$O/parity_margins.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^margin" | tail -12 > $O/tests.txt; tail -5 $O/tests.txt
timeout 600 python scripts/train_geometry_bench.py > $O/train_geometry.txt 2>&1; tail -12 $O/train_geometry.txt
timeout 600 python scripts/dsac_variant_bench.py > $O/dsac_variant.txt 2>&1; tail -12 $O/dsac_variant.txt
timeout 600 python bench.py --steps 40 --warmup 10 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.1f us/step %.3f Mhyp/s K2 %.1f us frac %.3f, per-image / kernel-only rate %.3f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['avg_launch_us'], d['roofline']['frac'], d['rates']['per_image_hyp_s']/d['rates']['kernel_only_k2_hyp_s']))"

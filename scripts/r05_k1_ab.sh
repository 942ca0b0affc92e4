#!/bin/bash
# K1 built without MachineLICM (250 registers, no scratch, two waves per SIMD: the default since round 5) against the round-4 build (331 registers incl. 75
# accumulation registers that hold hoisted libm constants, one wave per SIMD): scripts/k1_bench.py on one 640x480 frame, then the default bench step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05k1; mkdir -p $O
OLD=$PWD/dsac_amd/csrc/build/ab/libdsac_hip_k1licm.so
for rep in 1 2; do
  echo "== K1 default (no MachineLICM, 2 waves per SIMD)"; timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
  echo "== K1 round-4 build (hoisted constants, 1 wave per SIMD)"; DSAC_HIP_LIB=$OLD timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N"
done | tee $O/k1_licm_ab.txt
for m in new old new old; do
  if [ $m = old ]; then export DSAC_HIP_LIB=$OLD; else unset DSAC_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-driver --no-single-frame 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K1 $m: %.1f us/step %.3f Mhyp/s K2 %.1f us, per-image / kernel-only rate %.3f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['avg_launch_us'], d['rates']['per_image_hyp_s']/d['rates']['kernel_only_k2_hyp_s']))"
done | tee -a $O/k1_licm_ab.txt

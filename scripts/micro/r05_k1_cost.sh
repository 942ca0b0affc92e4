#!/bin/bash
# what each piece of K1's OpenCV arithmetic costs, standalone (scripts/k1_bench.py) and in the default bench step, and what it buys (sets / poses against the
# oracle): the build in the tree against pow -> cbrt, contraction off only inside the P3P solve (pragmas), both, and the orthonormal triad
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
export DSAC_DIAG_BRIEF=1
{
for L in default cbrt pragmaonly cbrt_pragmaonly triad; do
  if [ $L = default ]; then unset DSAC_HIP_LIB; else export DSAC_HIP_LIB=$PWD/dsac_amd/csrc/build/ab/libdsac_hip_k1$L.so; fi
  echo "== $L"
  timeout 300 python scripts/micro/r05_k1_accept_diag.py 2>&1 | python -c "
import sys,re
t=sys.stdin.read()
tot=re.findall(r'total differing (\d+)',t)
w=[float(x) for x in re.findall(r'^   poses on the \d+ identical sets: [\d.]+ % bit-equal, ([\d.]+) % within 1e-9',t,re.M)]
b=[float(x) for x in re.findall(r'^   poses on the \d+ identical sets: .*?([\d.]+) % beyond 1e-6',t,re.M)]
print('   sets differing %s of 27648; poses within 1e-9: mean %.2f %% min %.2f %%; beyond 1e-6: mean %.3f %% max %.3f %%' % (tot, sum(w)/len(w), min(w), sum(b)/len(b), max(b)))"
  timeout 300 python scripts/k1_bench.py 2>&1 | grep "K1 N= 4096\|K1 N=  256"
  for rep in 1 2; do timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-driver --no-single-frame 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   step %.1f us %.3f Mhyp/s K2 %.1f us, per-image / kernel-only rate %.4f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['avg_launch_us'], d['rates']['per_image_hyp_s']/d['rates']['kernel_only_k2_hyp_s']))"; done
done
} > $O/r05_k1_cost.txt 2>&1
cat $O/r05_k1_cost.txt

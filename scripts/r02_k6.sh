#!/bin/bash
# K6 alone under rocprofv3: B = 1 and B = 256 problems; the table goes to gpurun_out/r02/k6_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=$PWD/gpurun_out/r02; mkdir -p $O; TAG=${1:-base}
export TMPDIR=/tmp
{
for B in 1 256; do
  rm -rf /tmp/k6p; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k6p -o k6 --output-format csv -- python $OLDPWD/scripts/k6_bench.py $B 20 2>/dev/null | grep k_refine)
  python - <<PY
import csv, glob
f = glob.glob('/tmp/k6p/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if r['Kernel_Name'].startswith('dk::k_refine(')]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
h = len(d) // 2
print("  rocprofv3 k_refine B=$B: 40x40 %.1f us avg (min %.1f), 640x480 %.1f us avg (min %.1f), %d launches" % (sum(d[1:h]) / (h - 1), min(d[1:h]), sum(d[h + 1:]) / (h - 1), min(d[h + 1:]), len(d)))
PY
done
} 2>&1 | tee $O/k6_$TAG.txt

#!/bin/bash
# kernel timeline of the loop of single images (C++ program, one image per launch chain) in the three deferral modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); O=gpurun_out/r04si; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for d in 0 1 2; do
  rm -rf /tmp/si$d
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/si$d -o k -- $REPO/dsac_amd/host/test_ransac_softam -synth 64 -mw 640 -mh 480 -batch 1 -passes 4 -defer $d > /tmp/si$d.log 2>&1
  python - <<PY
import csv, glob, statistics as st
f = glob.glob("/tmp/si$d/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))), key=lambda t: t[1])
k2 = [(s, e) for n, s, e in rows if "k_reproject" in n][-101:]
lo = k2[0][0]
k1 = [(s, e) for n, s, e in rows if "k_sample" in n and s >= lo]
k6 = [(s, e) for n, s, e in rows if "k_refine" in n and s >= lo][:100]
k3 = [(s, e) for n, s, e in rows if "k_softmax" in n and s >= lo][:100]
per = [(b[0] - a[0]) / 1e3 for a, b in zip(k2[:-1], k2[1:])]
def ov(ks, big):
    tot = o = 0
    for s, e in ks:
        tot += e - s
        o += sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in big if e2 > s and s2 < e)
    return 100.0 * o / max(1, tot)
k6o = 0; tot6 = 0
for i, (s, e) in enumerate(k6):
    tot6 += e - s
    k6o += sum(max(0, min(e, e2) - max(s, s2)) for j, (s2, e2) in enumerate(k6) if j != i and e2 > s and s2 < e)
print("-defer $d: period (K2 start to K2 start) median %.1f us over %d images; K1 %.1f, K2 %.1f, K3 %.1f, K6 %.1f us (medians); of K6: %.0f %% under a K1 / K2, %.0f %% beside ANOTHER image's K6" % (
    st.median(per), len(per), st.median([(e - s) / 1e3 for s, e in k1]), st.median([(e - s) / 1e3 for s, e in k2]), st.median([(e - s) / 1e3 for s, e in k3]),
    st.median([(e - s) / 1e3 for s, e in k6]), ov(k6, k1 + k2), 100.0 * k6o / max(1, tot6)))
PY
done | tee $REPO/$O/single_image_timeline.txt

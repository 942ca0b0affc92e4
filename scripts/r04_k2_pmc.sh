#!/bin/bash
# HBM traffic of the default bench step's K2 launch from the PMC counters, written to profiles/k2_traffic.json by this script (the
# file used to be assembled by hand from the logs).  Two separate --pmc passes (WRITE_SIZE, FETCH_SIZE) with --kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; KB = 1024 B; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B).
# usage (on the GPU box): bash scripts/r04_k2_pmc.sh   -> gpurun_out/r04/k2_pmc.txt, gpurun_out/r04/k2_traffic.json (copy the latter to profiles/)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r04"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
K2F="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-frame --no-host-driver --event-stride 0 --prewarm-ms 0 $DSAC_PMC_EXTRA"
: > "$OUT/k2_pmc.txt"
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- $K2F > /tmp/pmc_$c.log 2>&1
  python - /tmp/pmc_$c/p_counter_collection.csv $c >> "$OUT/k2_pmc.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_reproject" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
done
cat "$OUT/k2_pmc.txt"
python - "$OUT/k2_pmc.txt" "$OUT/k2_traffic.json" <<'PY'
import json, re, sys
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"(WRITE_SIZE|FETCH_SIZE) (.+) n=(\d+) mean=([0-9.e+-]+)", line.strip())
    if m:
        rows.setdefault(m.group(2), {})[m.group(1)] = (float(m.group(4)), int(m.group(3)))
# the step's launch = the instantiation that writes error images AND sums (ERR = SOFT = true) and runs in every timed step: of the kernels that write a whole
# launch's error images, the one launched most often (since round 6 the line also times the other K2 forms a few times each)
big = [k for k in rows if rows[k].get("WRITE_SIZE", (0, 0))[0] > 1e6 and "FETCH_SIZE" in rows[k]]
name = max(big, key=lambda k: rows[k]["WRITE_SIZE"][1])
w, f = rows[name]["WRITE_SIZE"][0], rows[name]["FETCH_SIZE"][0]
form = "fast" if re.search(r", 0>$", name.strip()) else "exact"  # last template argument of k_reproject_st: 0 = fp32 transform, 1 / 2 = exact transform
out = {"N": 4096, "frames": 16, "P": 307200, "kernel": name, "form": form, "WRITE_SIZE_KB": w, "FETCH_SIZE_KB_raw": f,
       "hbm_bytes_per_launch": int(round((w + 2.0 * f) * 1024.0)), "launches_averaged": rows[name]["WRITE_SIZE"][1],
       "note": "separate rocprofv3 --pmc passes for WRITE_SIZE and FETCH_SIZE of `bench.py --steps 6 --warmup 2` (16 frames x 256 hypotheses per launch), "
               "--kernel-trace only; KB = 1024 B; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
       "source": "written by scripts/r04_k2_pmc.sh from gpurun_out/r04/k2_pmc.txt (committed as profiles/r04_k2_pmc.txt)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY

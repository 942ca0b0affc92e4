#!/usr/bin/env python3
"""Instruction mix of one gfx950 kernel, read from the built object (no hand-typed counts): the device code object is pulled out of the host object's
offload bundle (llvm-objdump --offloading), disassembled, and the VALU instructions of the named kernel are counted by issue class
   packed fp32 (v_pk_*) | quarter-rate transcendentals (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos _f32) | MFMA | every other VALU op.
Only the hot path is counted: the code up to the kernel's first s_endpgm (cold blocks are laid out behind it).
`units` = how many (16 hypotheses x 64 pixels) = 1024-pair units the fully unrolled kernel body holds (groups x chunks), so that the counts can be quoted
per 1024 pairs like DESIGN.md does.  Used by dsac_amd/csrc/Makefile for bench.py's soft_only.issue_model.
usage: isa_mix.py <host object> <mangled-name regex> <units> <out.json>"""
import json
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|legacy_f32)")


def main(obj, pattern, units, out):
    obj = os.path.abspath(obj)
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(obj))
        os.symlink(obj, base)
        subprocess.run([OBJDUMP, "--offloading", base], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co:
            raise SystemExit("no gfx950 code object in %s" % obj)
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    rx = re.compile(pattern)
    name, counts, inside = None, None, False
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if inside:
                break
            if rx.search(m.group(1)):
                name, inside = m.group(1), True
                counts = dict(packed_fp32=0, transcendental=0, mfma=0, plain_valu=0, salu=0, vmem=0, lds=0, other=0)
            continue
        if not inside:
            continue
        t = line.strip().split()
        if not t or t[0].startswith("//"):
            continue
        op = t[0]
        if op == "s_endpgm":
            # the hot path ends here: what the compiler lays out behind the first s_endpgm are its cold blocks (the exact redo of a chunk with a Z == 0
            # hit sits under __builtin_expect(..., 0) and is never executed on ordinary frames) -- they are not part of the issue price
            break
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            counts["mfma"] += 1
        elif op.startswith("v_pk_"):
            counts["packed_fp32"] += 1
        elif TRANS.match(op):
            counts["transcendental"] += 1
        elif op.startswith("v_") and op not in ("v_nop",):
            counts["plain_valu"] += 1
        elif op.startswith("s_"):
            counts["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            counts["vmem"] += 1
        elif op.startswith("ds_"):
            counts["lds"] += 1
        else:
            counts["other"] += 1
    if name is None:
        raise SystemExit("no kernel matches %r" % pattern)
    units = int(units)
    res = {"kernel": name, "units_of_1024_pairs": units, "counts": counts,
           "per_1024_pairs": {k: counts[k] / units for k in ("packed_fp32", "transcendental", "mfma", "plain_valu")},
           "source": "llvm-objdump -d of the gfx950 code object inside %s (scripts/isa_mix.py, run by dsac_amd/csrc/Makefile)" % os.path.basename(obj)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print("%s: %s" % (out, json.dumps(res["per_1024_pairs"])))


if __name__ == "__main__":
    main(*sys.argv[1:5])

#!/usr/bin/env python3
"""Instruction mix of the hottest LOOP of a gfx950 kernel, read from the built object: the kernel is disassembled, loops are found by their backward
branches, and the opcode histogram of the chosen loop body (by default the backward branch with the largest body that contains no other backward branch's
target... i.e. the innermost big loop) is printed by issue class:
   matrix core | packed fp32 | transcendental (quarter rate) | scalar fp32 arithmetic (fma / mul / add / sub / mac) | min / max / compare / select |
   moves (v_mov, v_accvgpr, readlane, permlane, swap, dpp-only moves) | conversions | integer / address VALU | LDS | VMEM | SALU / branches / waits
usage: isa_loop_mix.py <host object> <mangled-name regex> [pairs per loop trip] [out.txt]
K4 (round 6): k_score_backward_mfma<4,false,false,2>: one trip of the loop over 16-hypothesis groups = 16 hypotheses x 4 chunks x 16 pixels
per wave = 1 024 (hypothesis, pixel) pairs."""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|legacy_f32|f16)")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "matrix core"
    if op.startswith("v_pk_"):
        return "packed fp32"
    if TRANS.match(op):
        return "transcendental"
    if re.match(r"^v_(fma|fmac|mul|add|sub|mac|mad|fmaak|fmamk)_(f32|legacy_f32|f64|dx9_zero_f32)", op) or re.match(r"^v_(subrev|mul_legacy)_f32", op):
        return "scalar fp arithmetic"
    if re.match(r"^v_(min|max|med3|cmp|cmpx|cndmask|min3|max3)", op):
        return "min / max / compare / select"
    if re.match(r"^v_(mov|accvgpr|readlane|readfirstlane|writelane|permlane|swap|bfi|perm_b32|pack)", op):
        return "moves / lane exchange"
    if op.startswith("v_cvt") or op.startswith("v_rndne") or op.startswith("v_floor") or op.startswith("v_trunc") or op.startswith("v_ldexp") or op.startswith("v_frexp"):
        return "conversions"
    if op.startswith("v_") and op != "v_nop":
        return "integer / address VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    return "SALU / branch / wait / nop"


def main(obj, pattern, pairs=None, out=None):
    obj = os.path.abspath(obj)
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(obj))
        os.symlink(obj, base)
        subprocess.run([OBJDUMP, "--offloading", base], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    rx = re.compile(pattern)
    name, ins, inside = None, [], False
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if inside:
                break
            if rx.search(m.group(1)):
                name, inside = m.group(1), True
            continue
        if not inside:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    if not ins:
        raise SystemExit("kernel not found: %s" % pattern)
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    # backward branches: s_cbranch_* / s_branch whose target address (printed by objdump as <name+0xOFF>) lies before the branch
    base_addr = ins[0][0]
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
            if not m:
                continue
            tgt = base_addr + int(m.group(1), 16)
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
    if not loops:
        raise SystemExit("no loop found")
    # innermost loops = those that contain no other loop; take the one with the most matrix-core / VALU instructions
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    best = max(inner, key=lambda l: sum(1 for (_, op, _) in ins[l[0]:l[1] + 1] if op.startswith("v_")))
    body = ins[best[0]:best[1] + 1]
    classes, ops = {}, {}
    for _, op, _ in body:
        c = classify(op)
        classes[c] = classes.get(c, 0) + 1
        ops.setdefault(c, {})
        ops[c][op] = ops[c].get(op, 0) + 1
    lines = ["kernel %s" % name, "loop body: %d instructions (offsets 0x%x .. 0x%x); %d loops in the kernel, %d innermost" %
             (len(body), body[0][0] - base_addr, body[-1][0] - base_addr, len(loops), len(inner))]
    valu = sum(v for k, v in classes.items() if k not in ("LDS", "VMEM", "SALU / branch / wait / nop"))
    if pairs:
        lines.append("per trip: %d (hypothesis, pixel) pairs per wave = %.1f lane-pairs of 64; VALU + matrix-core instructions per trip %d = %.2f per 64-lane pair-slot "
                     "(x 64 lanes / pairs: %.2f instructions per pair)" % (pairs, pairs / 64.0, valu, valu / (pairs / 64.0), valu * 64.0 / pairs))
    for c in sorted(classes, key=lambda k: -classes[k]):
        top = ", ".join("%s %d" % kv for kv in sorted(ops[c].items(), key=lambda kv: -kv[1])[:8])
        lines.append("  %-34s %5d%s   %s" % (c, classes[c], ("  (%.2f per 1 024 pairs)" % (classes[c] * 1024.0 / pairs)) if pairs else "", top))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else None)

#!/usr/bin/env python3
"""Registers / scratch of the kernels in a built host object (llvm-readelf --notes of its gfx950 code object).
usage: kernel_regs.py <host object> <substring of the mangled name> ..."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main(obj, pats):
    obj = os.path.abspath(obj)
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(obj))
        os.symlink(obj, base)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", base], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    for b in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        m = re.search(r"\.name:\s+(\S+)", b)
        if not m or not all(p in m.group(1) for p in pats):
            continue
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        print("%s  agpr %s vgpr %s sgpr %s scratch %s lds %s" % (m.group(1), b.split()[0], get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                                                                 get("group_segment_fixed_size")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

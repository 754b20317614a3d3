#!/usr/bin/env python
"""Audit of csrc/gemm_lw.hip's gfx950 ISA (the rule for asm-owned registers): the accumulator file a[0:255] is named only inside the
kernel's inline asm; nothing is spilled; prints per-kernel instruction mixes of the K loop. Usage: lw_audit.py [file.s]"""
import re
import sys


def kernels(text):
    lines = text.split("\n")
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\d+gemm_bf16_lw_kernel\w+):", l)
        if m:
            e = next(k for k in range(i, len(lines)) if "s_endpgm" in lines[k])
            out.append((m.group(1), lines[i:e + 1]))
    return out


def audit(body):
    inasm, bad, n_mfma = False, [], 0
    for l in body:
        if "#ASMSTART" in l:
            inasm = True
        elif "#ASMEND" in l:
            inasm = False
        elif not inasm:
            t = l.split(";")[0]
            if re.search(r"\ba\[?\d+", t) or "accvgpr" in t:
                bad.append(l.strip())
        if "v_mfma" in l and inasm:
            n_mfma += 1
    return n_mfma, bad


if __name__ == "__main__":
    text = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/asm/gemm_lw.s").read()
    for name, body in kernels(text):
        n, bad = audit(body)
        scratch = [l for l in body if "scratch_" in l.split(";")[0]]
        print(f"{name[:40]:40s} lines {len(body):6d} asm MFMAs {n:4d} compiler-touched AGPR lines {len(bad)} scratch {len(scratch)}")
        for b in bad[:5]:
            print("   ", b)

#!/usr/bin/env python
"""Static instruction mix of every kernel in a hipcc --cuda-device-only -S assembly file (VALU / MFMA / SALU / DS / VMEM,
VGPR count, scratch): the first check of an epilogue change before a GPU run.   python tools/asm_mix.py /tmp/gemm_pp.s"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
name, cnt, out = None, None, []
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, cnt = m.group(1), dict(valu=0, mfma=0, salu=0, ds=0, vmem=0)
        continue
    if name is None:
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", ln)
    if m:
        op = m.group(1)
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("v_"):
            cnt["valu"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
        elif op.startswith("ds_"):
            cnt["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cnt["vmem"] += 1
        if op == "s_endpgm":
            out.append((name, cnt))
            name = None
vg = dict(re.findall(r"\.name:\s+(_Z\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", "\n".join(txt)))
for n, c in out:
    short = re.sub(r"^_Z\d+", "", n)[:60]
    print(f"{short:62s} valu={c['valu']:5d} mfma={c['mfma']:4d} salu={c['salu']:5d} ds={c['ds']:4d} vmem={c['vmem']:4d} vgpr={vg.get(n, '?')}")

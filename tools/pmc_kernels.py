#!/usr/bin/env python
"""Fold a rocprofv3 `--pmc` counter_collection CSV into per-kernel sums and the derived figures the GEMM / attention
analysis needs (gfx950, 256 CUs x 4 SIMDs):

    python tools/pmc_kernels.py <counter_collection.csv> [<kernel_trace.csv>] > summary.json

  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCD instances (a 2.0 ms launch at 1.65 GHz reads 8 x 3.3e6), and
  SQ_VALU_MFMA_BUSY_CYCLES exactly 16 per v_mfma_f32_16x16x32_bf16 / 32 per 32x32x16 (checked against the flop count):
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)    busy matrix-pipe cycles per SIMD-cycle
  eff_clock_GHz  = GRBM_GUI_ACTIVE / 8 / kernel duration                             (needs the kernel trace of the same run)
  wait / issue   = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2_hit         = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main(pmc_csv, trace_csv=None):
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for row in csv.DictReader(open(pmc_csv, newline="")):
        k = short(row["Kernel_Name"])
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    dur = defaultdict(float)
    ndur = defaultdict(int)
    if trace_csv:
        for row in csv.DictReader(open(trace_csv, newline="")):
            k = short(row["Kernel_Name"])
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            ndur[k] += 1
    out = {}
    for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        d = {"launches": len(launches[k]), "counters": dict(c)}
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0)
        if gui and dur.get(k):
            d["duration_us_per_launch"] = dur[k] / ndur[k] / 1e3
            d["eff_clock_GHz"] = gui / 8.0 / dur[k]
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    d[n.lower() + "_frac"] = c[n] / wc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        out[k] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean over dispatches):
duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA pipe busy, wait breakdown, HBM bytes
(FETCH_SIZE x 2 on gfx950 for wide streaming reads, MI355X_MICROARCH.md section HBM).
Usage: summarize_pmc.py <counter_collection.csv> [kernel substring]"""
import collections
import csv
import sys

path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "corrla"
disp = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if filt not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0], "grid": int(r["Grid_Size"]),
                                           "wg": int(r["Workgroup_Size"]),
                                           "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in disp.values():
    a = agg.setdefault((d["name"], d["grid"]), collections.defaultdict(list))
    for k, v in d.items():
        if k not in ("name",):
            a[k].append(v)
for (name, grid), a in agg.items():
    m = {k: sum(v) / len(v) for k, v in a.items()}
    n = len(a["dur_ns"])
    line = f"{name} grid={grid} n={n} dur={m['dur_ns'] / 1e3:.1f}us"
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += f" clock={cyc / m['dur_ns']:.2f}GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            nsimd = 4 * 256
            line += f" mfma_busy={100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / nsimd / cyc:.1f}%"
    if "SQ_WAVE_CYCLES" in m:
        for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
            if c in m:
                line += f" {lab}={100 * m[c] / m['SQ_WAVE_CYCLES']:.1f}%"
    if "FETCH_SIZE" in m:
        line += f" FETCH_SIZE={m['FETCH_SIZE']:.0f}KB (x2 corrected: {m['FETCH_SIZE'] * 2 * 1024 / 1e9:.3f} GB)"
    if "WRITE_SIZE" in m:
        line += f" WRITE_SIZE={m['WRITE_SIZE']:.0f}KB"
    for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "TCC_HIT_sum",
              "TCC_MISS_sum"):
        if c in m:
            line += f" {c}={m[c]:.4g}"
    print(line)

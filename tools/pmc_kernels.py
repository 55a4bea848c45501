#!/usr/bin/env python3
"""Per-kernel-class sums of a `rocprofv3 --kernel-trace --pmc ...` pass: python tools/pmc_kernels.py <dir> [min_duration_us].
Prints, for every kernel class, the counter totals over its dispatches and a few ratios (MFMA-busy fraction, wave-cycle split)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))


def cls(n):
    for key in ("wgrad_x6", "wino2d_x6", "wino2d", "wgrad_wino", "wgrad_f32", "igemm_wino", "igemm_f32", "attn_", "gn_"):
        if key in n:
            return key
    return "other"


disp = collections.defaultdict(dict)
for r in rows:
    e = disp[r["Dispatch_Id"]]
    e["name"] = r["Kernel_Name"]
    e[r["Counter_Name"]] = float(r["Counter_Value"])
    e["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for e in disp.values():
    if e["dur"] < min_ns:
        continue
    a = agg[cls(e["name"])]
    for k, v in e.items():
        if isinstance(v, float):
            a[k] += v
    a["dur_ns"] += e["dur"]; a["n"] += 1
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur_ns"]):
    out = {c: v for c, v in a.items()}
    line = f"{k:12s} n={int(a['n']):5d} ms={a['dur_ns'] / 1e6:8.2f}"
    if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        line += f"  mfma_busy={a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}"
    if a.get("SQ_WAVE_CYCLES"):
        w = a["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM"):
            if c in a:
                line += f"  {c[3:]}={a[c] / w:.3f}"
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
              "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CU_CYCLES"):
        if c in a:
            line += f"  {c[3:]}={a[c]:.3e}"
    print(line)

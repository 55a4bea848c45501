#!/usr/bin/env python3
"""Turns the rocprofv3 outputs under gpurun_out/ (kernel stats + three --pmc passes) into the committed
summaries under profiles/.  Usage: python tools/summarize_profiles.py <stats_dir> <pmc_prefix> <tag> <steps> [<short_stats_dir> <short_steps>]
With the optional short trace (same command, fewer steps) the per-class table is the STEADY-STATE step: (long - short) / (steps
difference), which removes model construction, first-use weight packing and other one-time launches from the per-step figures."""
import collections
import csv
import glob
import os
import json
import sys

stats_dir, pmc_prefix, tag, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
short_dir, short_steps = (sys.argv[5], int(sys.argv[6])) if len(sys.argv) > 6 else (None, 0)


def load(pattern):
    return list(csv.DictReader(open(max(glob.glob(pattern), key=os.path.getmtime))))      # (the newest run, if several were merged)


def cls(n):
    if "wgrad_wino_kernel<2>" in n:
        return "wgrad_wino2d"
    if "wino2d_x6" in n:
        return "wino2d_x6"
    if "wgrad_x6" in n:
        return "wgrad_x6"
    if "gemm_x6" in n:
        return "gemm_x6"
    if "igemm_wino2d" in n:
        return "wino2d"
    return ("wgrad_wino" if "wgrad_wino" in n else "wgrad" if "wgrad" in n else "wino" if "wino" in n else "igemm" if "igemm" in n else "attn" if "attn" in n else "gn" if "::gn_" in n
            else "other")


rows = load(stats_dir + "/*/*_kernel_stats.csv")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")])
by = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    by[cls(r["Name"])][0] += float(r["TotalDurationNs"]); by[cls(r["Name"])][1] += int(r["Calls"])
nsteps = steps
if short_dir:
    for r in load(short_dir + "/*/*_kernel_stats.csv"):
        by[cls(r["Name"])][0] -= float(r["TotalDurationNs"]); by[cls(r["Name"])][1] -= int(r["Calls"])
    nsteps = steps - short_steps
tot_ss = sum(v[0] for v in by.values())
md = [f"# rocprofv3 --kernel-trace --stats ({tag})", "",
      "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps %d --warmup 1 --profile-only` "
      "(bs=128 fp32, full 216M-param CIFAR-10 UNet; %d training steps in the trace)." % (steps - 1, steps), ""]
if short_dir:
    md += [f"Steady state: the same command with {short_steps} steps is subtracted, so model construction, first-use weight packing and "
           f"other one-time launches are out of the per-step figures below ({nsteps} steps remain).", ""]
md += [f"Kernel time per steady-state step: **{tot_ss / 1e6 / nsteps:.1f} ms**, {sum(v[1] for v in by.values()) / nsteps:.0f} launches "
       f"(whole trace: {tot / 1e6:.1f} ms).  The sum exceeds the wall-clock step of an untraced run because the tracer adds a "
       "fixed cost to every dispatch, and because the traced run keeps everything on ONE stream (ADM_SIDE_WGRAD=0 ADM_BRANCH_STREAM=0: "
       "true per-kernel durations) while the untraced step overlaps the weight gradients and the second decoder with the main chain.", "",
       "| class | launches/step | ms/step | avg launch us |", "|---|---|---|---|"]
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    if n > 0:
        md.append(f"| {k} | {n / nsteps:.0f} | {t / 1e6 / nsteps:.2f} | {t / 1e3 / n:.1f} |")
per = collections.OrderedDict()
for r in rows:
    per[r["Name"]] = [float(r["TotalDurationNs"]), int(r["Calls"])]
if short_dir:
    for r in load(short_dir + "/*/*_kernel_stats.csv"):
        if r["Name"] in per:
            per[r["Name"]][0] -= float(r["TotalDurationNs"]); per[r["Name"]][1] -= int(r["Calls"])
md += ["", "| kernel | launches/step | ms/step | avg us | % of kernel time |", "|---|---|---|---|---|"]
for name, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:36]:
    if n > 0:
        md.append(f"| `{name[:84]}` | {n / nsteps:.1f} | {t / 1e6 / nsteps:.2f} | {t / 1e3 / n:.1f} | {100 * t / tot_ss:.2f} |")

# ---- PMC passes (each counter group collected in its own run, as MI355X_MICROARCH.md prescribes)
cc = load(pmc_prefix + "GRBM_GUI_ACTIVE/*/*_counter_collection.csv")
disp = collections.defaultdict(dict)
for r in cc:
    d = disp[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"]; d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    a = agg[cls(d["name"])]
    if d["dur"] >= 200000:       # the effective-clock estimate is only valid for dispatches >= ~0.2-0.3 ms
        for k, v in d.items():
            if isinstance(v, float):
                a[k] += v
        a["dur"] += d["dur"]; a["n"] += 1
out = {}
for k, a in agg.items():
    if a.get("GRBM_GUI_ACTIVE"):
        e = {"dispatches_over_0.2ms": int(a["n"]),
             "effective_clock_GHz": round(a["GRBM_GUI_ACTIVE"] / 8 / (a["dur"] * 1e-9) / 1e9, 3)}
        if a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["mfma_busy_fraction"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
        out[k] = e
f = collections.defaultdict(lambda: [0.0, 0]); w = collections.defaultdict(lambda: [0.0, 0])
for r in load(pmc_prefix + "FETCH_SIZE/*/*_counter_collection.csv"):
    f[cls(r["Kernel_Name"])][0] += float(r["Counter_Value"]); f[cls(r["Kernel_Name"])][1] += 1
for r in load(pmc_prefix + "WRITE_SIZE/*/*_counter_collection.csv"):
    w[cls(r["Kernel_Name"])][0] += float(r["Counter_Value"]); w[cls(r["Kernel_Name"])][1] += 1
for k in f:
    e = out.setdefault(k, {})
    n = f[k][1]
    fetch = f[k][0] * 1024 * 2      # gfx950: FETCH_SIZE (KB) reports HALF the bytes of wide coalesced reads -> x2
    write = w[k][0] * 1024          # WRITE_SIZE (KB) is exact for 16-byte streaming stores
    e.update({"launches_in_pmc_run": n, "fetch_bytes_per_launch_x2_corrected": round(fetch / n),
              "write_bytes_per_launch": round(write / n), "traffic_bytes_per_launch": round((fetch + write) / n)})
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
md += ["", "## PMC (separate `rocprofv3 --kernel-trace --pmc <group>` passes: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE "
       "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES)", "",
       "FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); it counts "
       "L2 misses served by the Infinity Cache as well as HBM.  `mfma_busy_fraction` = SQ_VALU_MFMA_BUSY_CYCLES / "
       "(GRBM_GUI_ACTIVE/8 x 1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / 8 / dispatch time.", "", "```json",
       json.dumps(out, indent=1), "```"]
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(md) + "\n")
print("\n".join(md[:14])); print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Idle-time analysis of a rocprofv3 --kernel-trace CSV: over the last N steps of the trace (split at the AdamW launches), the wall
time of a step, the union of the kernel intervals (GPU busy), the idle remainder and where the idle gaps sit (by the kernel that
FOLLOWS the gap).  Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [steps]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    if len(marks) < nsteps + 1:
        print("not enough steps in the trace", len(marks)); return
    lo, hi = marks[-nsteps - 1] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy, cur_end, gaps = 0, seg[0][0], defaultdict(lambda: [0, 0.0])
    sum_dur = 0
    streams = defaultdict(float)
    for s, e, name, q in seg:
        sum_dur += e - s
        streams[q] += (e - s) / 1e6
        if s > cur_end:
            g = (s - cur_end) / 1e3
            key = name.split("(")[0][-48:]
            gaps[key][0] += 1; gaps[key][1] += g
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    busy /= 1e6
    print(f"{nsteps} steps: wall {wall / nsteps:.2f} ms/step, GPU busy (union of kernels) {busy / nsteps:.2f}, idle {(wall - busy) / nsteps:.2f}, "
          f"sum of kernel durations {sum_dur / 1e6 / nsteps:.2f}, launches {len(seg) / nsteps:.0f}")
    print("kernel time by stream/queue (ms/step):", {k: round(v / nsteps, 2) for k, v in streams.items()})
    hist = defaultdict(lambda: [0, 0.0])
    for k, (n, t) in gaps.items():
        pass
    print("idle gaps by the kernel that follows them (us/step, count/step):")
    for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / nsteps:9.1f} us  {n / nsteps:7.1f}  {k}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive dispatches from a rocprofv3 --kernel-trace csv.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [name filter]"""
import csv
import glob
import statistics
import sys

f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
by = {}
for i, (s, e, n) in enumerate(rows):
    short = n[:70]
    d = by.setdefault(short, {"dur": [], "gap_before": []})
    d["dur"].append((e - s) / 1e3)
    if i:
        d["gap_before"].append((s - rows[i - 1][1]) / 1e3)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1]["dur"])):
    if flt and flt not in n:
        continue
    g = d["gap_before"] or [0]
    print(f"{n:70s} n={len(d['dur']):6d} dur med {statistics.median(d['dur']):8.2f} mean {statistics.mean(d['dur']):8.2f} us | "
          f"gap before: med {statistics.median(g):6.2f} mean {statistics.mean(g):6.2f} p90 {sorted(g)[int(0.9 * (len(g) - 1))]:6.2f} us")

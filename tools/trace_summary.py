#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace csv directory: per kernel (and per grid size for the panel
kernels) count, mean and total duration, plus the time span and the gaps of the LAST fit in the
trace.   python tools/trace_summary.py DIR"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last fit: from the last fit_prologue_kernel on
starts = [i for i, r in enumerate(rows) if "fit_prologue" in r["Kernel_Name"]]
last = rows[starts[-1]:] if starts else rows
agg = defaultdict(lambda: [0, 0.0])
for r in last:
    name = r["Kernel_Name"].split("(")[0][:60]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = name
    if "panel" in name or "pivot" in name:
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "256")) or 256), 1)
        key = "%s[grid %s]" % (name, "1" if g <= 1 else ("<=64" if g <= 64 else ("<=256" if g <= 256 else ">256")))
    agg[key][0] += 1
    agg[key][1] += dur
t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
print("last fit: %d launches, span %.1f us, sum of kernel durations %.1f us" % (len(last), (t1 - t0) / 1e3, sum(v[1] for v in agg.values())))
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-72s n=%4d  mean %8.2f us  total %9.1f us" % (k, n, tot / n, tot))

#!/usr/bin/env python3
"""Durations of one kernel (substring match) in launch order, from a rocprofv3 kernel_trace.csv."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "calls; us:", " ".join(f"{x:.0f}" for x in d[::max(1, len(d) // 60)]))
g = {}
for r, x in zip(rows, d):
    k = (r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("Workgroup_Size_X", "?"))
    g.setdefault(k, []).append(x)
for k, v in g.items():
    print("grid", k, "calls", len(v), "avg us", round(sum(v) / len(v), 1), "min", round(min(v), 1), "max", round(max(v), 1))

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace output directory per kernel AND per dispatch shape (grid, workgroup, LDS): count,
average / min / max duration in us.  usage: kstats.py <dir> [substring filter]"""
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        key = (name[:110], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':110s} {'grid':>9s} {'wg':>5s} {'lds':>7s} {'n':>5s} {'avg us':>9s} {'min':>9s} {'max':>9s} {'share':>6s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:110s} {k[1]:>9s} {k[2]:>5s} {k[3]:>7s} {len(v):5d} {sum(v) / len(v):9.1f} {min(v):9.1f} {max(v):9.1f} {100 * sum(v) / tot:5.1f}%")

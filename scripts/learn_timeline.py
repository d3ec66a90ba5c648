#!/usr/bin/env python3
"""One learn step in launch order from a rocprofv3 kernel trace of scripts/learn_prof.py: start, gap to the previous kernel, duration.
usage: learn_timeline.py <trace dir>   (the last period between two k_td_loss launches)"""
import csv, glob, re, sys
fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=lambda f: -len(open(f).read()))
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_td_loss" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0, prev, tot = int(rows[a]["Start_Timestamp"]), None, 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:84]
    print(f"{(s - t0) / 1e3:8.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f} dur {(e - s) / 1e3:7.1f}  {n}")
    prev = e
    tot += (e - s) / 1e3
print("sum of kernels us", round(tot, 1), "period us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)

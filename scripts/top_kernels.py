#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f, "total kernel ms", round(tot / 1e6, 2))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>7s} {float(r["TotalDurationNs"]) / 1e6:9.2f} ms {r["Percentage"]:>6s}%  avg {float(r["AverageNs"]) / 1e3:8.1f} us')

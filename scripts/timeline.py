#!/usr/bin/env python3
"""One period of a kernel trace in launch order: timeline.py <rocprofv3 dir> <marker kernel substring> [period index from the end]
Prints every kernel between two consecutive launches of the marker (start, duration, gap to the previous kernel) and the sums."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
marker = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = starts[-back - 1], starts[-back]
t0 = prev = int(rows[a]["Start_Timestamp"])
busy = gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:80]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {name}")
    busy += e - s
    gaps += max(0, s - prev)
    prev = e
print(f"{b - a} kernels, busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, period {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")

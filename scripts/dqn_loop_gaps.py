#!/usr/bin/env python3
"""Device time vs wall of the DDQN loop from a rocprofv3 kernel trace of scripts/dqn_loop_prof.py: busy time, idle gaps > 5 us (where, how long).
usage: dqn_loop_gaps.py <trace dir>"""
import csv, glob, re, sys
fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=lambda f: -len(open(f).read()))
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the second half: past the warm-up
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy, prev, gaps = 0, None, {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev is not None and s - prev > 5000:
        n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:50]
        g = gaps.setdefault(n, [0, 0])
        g[0] += 1
        g[1] += s - prev
    prev = max(prev or e, e)
print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms: device busy {busy / 1e6:.2f} ms = {busy / (t1 - t0):.3f}")
for n, (c, ns) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  idle before {n:52s} {c:4d} times, {ns / 1e3:8.1f} us in all")

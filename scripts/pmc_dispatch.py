#!/usr/bin/env python3
"""HBM traffic PER DISPATCH from three runs of the same deterministic program: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE
and --kernel-trace.  Dispatch i of a kernel in one run is dispatch i in the others, so every dispatch gets ITS bytes and ITS
duration (the folded averages divided split byte groups by an unsplit mean duration).  Dispatches are then
grouped by duration class (more than 1.25x apart = different work: warm-up / timed / resident launches of one kernel).
Bytes follow MI355X_MICROARCH.md: counters are KiB; FETCH_SIZE x 2 on gfx950 for wide streaming reads.
usage: pmc_dispatch.py <fetch dir> <write dir> <trace dir> <out.json> <kernel substr> [steps per launch for the longest class]"""
import csv, glob, json, sys, collections
fd, wd, td, out_json, filt = sys.argv[1:6]

def counter(d, name):
    v = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and filt in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            v[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return v

def trace(d):
    v = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            v[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return v

F, W, T = counter(fd, "FETCH_SIZE"), counter(wd, "WRITE_SIZE"), trace(td)
out = {"note": "per-dispatch: bytes from the PMC passes, duration from the kernel-trace pass of the same program; FETCH_SIZE x 2 (gfx950)",
       "kernels": {}}
for k in T:
    n = min(len(F.get(k, [])), len(W.get(k, [])), len(T[k]))
    if n == 0:
        continue
    disp = [{"us": T[k][i], "fetch": F[k][i] * 2048.0, "write": W[k][i] * 1024.0} for i in range(n)]
    def split(items, key, ratio):
        items = sorted(items, key=key)
        groups, cur = [], [items[0]]
        for d in items[1:]:
            if key(d) > ratio * max(key(cur[0]), 1e-9):
                groups.append(cur)
                cur = [d]
            else:
                cur.append(d)
        groups.append(cur)
        return groups
    # like dispatches = similar fetch volume (a resident and a re-reading launch of one kernel differ 10x there) AND
    # similar duration (a 5-step warm-up and a 20-step timed launch)
    classes = [c for grp in split(disp, lambda d: d["fetch"], 1.3) for c in split(grp, lambda d: d["us"], 1.25)]
    rows = []
    for c in classes:
        us = sum(d["us"] for d in c) / len(c)
        fb, wb = sum(d["fetch"] for d in c) / len(c), sum(d["write"] for d in c) / len(c)
        rows.append({"dispatches": len(c), "avg_us": us, "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                     "hbm_GBps": (fb + wb) / us / 1e3})
    out["kernels"][k[:100]] = {"dispatches_matched": n, "dispatch_counts": [len(F.get(k, [])), len(W.get(k, [])), len(T[k])],
                               "classes": rows}
json.dump(out, open(out_json, "w"), indent=1)
print(json.dumps(out, indent=1))

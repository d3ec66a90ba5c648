#!/usr/bin/env python3
"""Fold rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) and a kernel-trace stats file
into profiles/<tag>_summary.json.  HBM bytes follow MI355X_MICROARCH.md §HBM: counters are KiB,
and on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so
it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv, glob, json, statistics, sys

def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in vals.items()}

def main(fetch_dir, write_dir, stats_csv, out_json, note, steps_per_launch=1):
    f = per_kernel(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
    w = per_kernel(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
    stats = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
    out = {"note": note, "steps_per_launch": int(steps_per_launch), "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if "k_tile" not in k and "k_obs" not in k:
            continue
        fb = f.get(k, 0.0) * 1024 * 2      # gfx950 correction
        wb = w.get(k, 0.0) * 1024
        s = stats.get(k)
        out["kernels"][k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                             "hbm_bytes_per_launch": fb + wb,
                             "avg_ns": float(s["AverageNs"]) if s else None, "calls": int(s["Calls"]) if s else None}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:7])

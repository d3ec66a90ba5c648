#!/usr/bin/env python3
"""Fold rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) and a kernel-trace stats file
into profiles/<tag>_summary.json.  HBM bytes follow MI355X_MICROARCH.md §HBM: counters are KiB,
and on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so
it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.

usage: pmc_summary.py <fetch dir> <write dir> <kernel_stats.csv> <out.json> <note> [steps_per_launch] [name filter,...]
Per kernel the MEDIAN dispatch is reported, and for the persistent rollout kernel the dispatches are split by
their resident / non-resident variant when both occur (they are told apart by their write volume)."""
import csv, glob, json, statistics, sys

def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return vals

def main(fetch_dir, write_dir, stats_csv, out_json, note, steps_per_launch=1, filt="k_tile,k_obs"):
    f = per_kernel(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
    w = per_kernel(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
    stats = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
    out = {"note": note, "steps_per_launch": int(steps_per_launch), "kernels": {}}
    keys = [k.strip() for k in filt.split(",") if k.strip()]
    # a kernel launched in two variants by the same program (rollout with / without resident boards; whole batch /
    # half batch): split its dispatches by the fetch volume — the two passes run the same program, so dispatch i of
    # the write pass is dispatch i of the fetch pass
    split = {}
    for k in list(f):
        fv = f[k]
        if len(fv) >= 4 and max(fv) > 1.6 * min(fv) and len(w.get(k, [])) == len(fv):
            thr = (max(fv) + min(fv)) / 2
            big = [i for i, v in enumerate(fv) if v >= thr]
            small = [i for i, v in enumerate(fv) if v < thr]
            wv = w[k]
            split[k + " [larger-fetch dispatches]"] = ([fv[i] for i in big], [wv[i] for i in big])
            split[k + " [smaller-fetch dispatches]"] = ([fv[i] for i in small], [wv[i] for i in small])
    for k2, (fv2, wv2) in split.items():
        f[k2], w[k2] = fv2, wv2
    for k in sorted(set(f) | set(w)):
        if keys and not any(x in k for x in keys):
            continue
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fb = statistics.median(fv) * 1024 * 2      # gfx950 correction
        wb = statistics.median(wv) * 1024
        s = stats.get(k.split(" [")[0])
        out["kernels"][k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                             "hbm_bytes_per_launch": fb + wb, "dispatches_fetch_pass": len(fv), "dispatches_write_pass": len(wv),
                             "fetch_min_max": [min(fv) * 2048, max(fv) * 2048], "write_min_max": [min(wv) * 1024, max(wv) * 1024],
                             "avg_ns": float(s["AverageNs"]) if s else None, "calls": int(s["Calls"]) if s else None}
        if s and float(s["AverageNs"]) > 0:
            out["kernels"][k]["hbm_GBps_at_avg_duration"] = (fb + wb) / float(s["AverageNs"])
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:8])

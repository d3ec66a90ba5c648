#!/usr/bin/env python3
"""HBM rate of the CNN's hand-written kernels inside the DQN loop: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
runs of `bench.py --dqn --steps 6 --repeats 1`) + the kernel-trace stats of the same command -> profiles/<tag>.json.
Counters are KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  The evidence BASELINE.json's
north_star asks for ("is the CNN memory-bound?"): bytes per dispatch / average duration against the 8 TB/s peak.

usage: dqn_hbm_summary.py <fetch dir> <write dir> <kernel_stats.csv> <out.json>"""
import csv, glob, json, sys


def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return vals


def short(name):
    for tag in ("k_conv3x3_f16", "k_conv3x3", "k_wgrad_reduce", "k_wgrad", "k_gemm_f16x3", "k_pool_split12", "k_bias_mish_bwd",
                "k_split_weights", "igemm_wrw", "igemm_bwd", "igemm_fwd", "batched_transpose"):
        if tag in name:
            return tag + (name[name.index(tag) + len(tag):][:14] if tag.startswith("k_conv3x3_f16") or tag == "k_wgrad" else "")
    return None


def main(fetch_dir, write_dir, stats_csv, out_json):
    f = per_kernel(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
    w = per_kernel(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
    stats = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
    out = {"note": "per dispatch, averaged over the dispatches of `bench.py --dqn --steps 6 --repeats 1` (policy forwards at "
                   "8 192 observations, learner / target forwards and gradients at 4 096); hbm_GBps = (2 x FETCH_SIZE + "
                   "WRITE_SIZE) KiB / average duration of the kernel-trace pass; peak 8 000", "kernels": {}}
    for name in sorted(set(f) | set(w)):
        tag = short(name)
        if tag is None or name not in stats:
            continue
        fb = 2 * 1024 * sum(f.get(name, [0])) / max(1, len(f.get(name, [0])))
        wb = 1024 * sum(w.get(name, [0])) / max(1, len(w.get(name, [0])))
        ns = float(stats[name]["AverageNs"])
        e = out["kernels"].setdefault(tag, {"dispatches": 0, "fetch_MB": 0.0, "write_MB": 0.0, "avg_us": 0.0, "_n": 0})
        n = int(stats[name]["Calls"])
        e["dispatches"] += n
        e["fetch_MB"] += fb / 1e6 * n
        e["write_MB"] += wb / 1e6 * n
        e["avg_us"] += ns / 1e3 * n
    for tag, e in out["kernels"].items():
        n = e["dispatches"]
        e["fetch_MB"], e["write_MB"], e["avg_us"] = round(e["fetch_MB"] / n, 2), round(e["write_MB"] / n, 2), round(e["avg_us"] / n, 1)
        e["hbm_GBps"] = round((e["fetch_MB"] + e["write_MB"]) / e["avg_us"] * 1e3, 1)
        e["frac_of_8TBps"] = round(e["hbm_GBps"] / 8000, 3)
        del e["_n"]
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])

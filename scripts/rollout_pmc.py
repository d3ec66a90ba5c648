#!/usr/bin/env python3
"""profiles/rNN_rollout_pmc.json — the record bench.py::pmc_traffic quotes — from three rocprofv3 runs of the SAME program
(`bench.py --only-rollout [--rollout-variant resident] --steps K --warmup W`: a warm-up launch and the timed launches of the
persistent rollout kernel, nothing else): --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes:
the counters do not fit one pass) and --kernel-trace (durations).  Dispatch i of the kernel in one run is dispatch i in the
others; the warm-up launch (a different step count) is dropped by its position, not by a duration threshold.  Units follow the
guide: the counters are KiB; FETCH_SIZE x 2 on gfx950 for wide streaming reads; WRITE_SIZE as it is.
usage: rollout_pmc.py <out.json> <round> --record <variant> <steps_per_launch> <warmup_launches> <fetch dir> <write dir> <trace dir> [--record ...]"""
import csv, glob, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    h = hashlib.sha256()
    for f in ("tron_env.hip", "tron_device.hpp"):
        h.update(open(os.path.join(ROOT, "deep-q-learning_tron_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def counter(d, name, filt):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and filt in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def trace(d, filt):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]


out_json, rnd = sys.argv[1], int(sys.argv[2])
args = sys.argv[3:]
records = []
while args:
    assert args[0] == "--record", args
    variant, spl, warm, fd, wd, td = args[1], int(args[2]), int(args[3]), args[4], args[5], args[6]
    args = args[7:]
    kern = "k_obs_roll"
    F, W, T = counter(fd, "FETCH_SIZE", kern)[warm:], counter(wd, "WRITE_SIZE", kern)[warm:], trace(td, kern)[warm:]
    n = min(len(F), len(W), len(T))
    assert n > 0, (variant, len(F), len(W), len(T))
    fb = sum(F[:n]) / n * 2048.0                      # KiB -> bytes, x 2 (gfx950: wide streaming reads are tallied at half)
    wb = sum(W[:n]) / n * 1024.0
    us = sum(T[:n]) / n
    records.append({"kernel": kern, "variant": variant, "envs": 65536, "width": 24, "mode": "none", "obs": "codes",
                    "steps_per_launch": spl, "dispatches": n, "dispatch_counts_fetch_write_trace": [len(F), len(W), len(T)],
                    "avg_us": us, "min_max_us": [min(T[:n]), max(T[:n])],
                    "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                    "hbm_bytes_per_step": (fb + wb) / spl, "hbm_GBps_at_avg_duration": (fb + wb) / us / 1e3,
                    "alg_bytes_per_launch": (2060 if variant == "plain" else 2 * 676 + 32) * 65536 * spl,
                    "alg_GBps_at_avg_duration": (2060 if variant == "plain" else 2 * 676 + 32) * 65536 * spl / us / 1e3})
json.dump({"round": rnd, "env_kernel_source_sha16": source_sha(),
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --kernel-trace, three runs of `python3 bench.py --only-rollout "
                   "[--rollout-variant resident] --no-cpu-baseline --steps K --warmup W --repeats R`: every dispatch past the warm-up "
                   "launches is the same launch; FETCH_SIZE x 2 (gfx950), counters in KiB",
           "records": records}, open(out_json, "w"), indent=1)
print(json.dumps(records, indent=1))

#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace output directory: rows per (kernel, grid, workgroup, LDS) AND per
duration class (dispatches of one shape whose durations differ by more than 1.25x are different work — a 5-step warm-up
and a 20-step timed launch of the persistent rollout share a grid), so that every average is over like dispatches.
With --batch B the weight-stationary convolutions get their FLOPs (2 B S^2 9 cin cout, parsed from the template
arguments) and TFLOP/s (f32-equivalent).  usage: kstats.py <dir> [--filter substr] [--batch B] [--json out.json]"""
import argparse, csv, glob, json, re, sys, collections
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--filter", default=None)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--json", default=None)
ap.add_argument("--skip-first", type=int, default=0, help="drop the first N dispatches of every kernel (warm-up)")
a = ap.parse_args()
rows = collections.defaultdict(list)
for f in glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if a.filter and a.filter not in name:
            continue
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
               r.get("LDS_Block_Size", "?"))
        rows[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out = []
for key, v in rows.items():
    v.sort()
    d = [x[1] for x in v][a.skip_first:]
    if not d:
        continue
    classes, cur = [], [sorted(d)[0]]
    for x in sorted(d)[1:]:
        if x > 1.25 * cur[0]:
            classes.append(cur)
            cur = [x]
        else:
            cur.append(x)
    classes.append(cur)
    for c in classes:
        row = {"kernel": key[0], "grid": key[1], "workgroup": key[2], "lds": key[3], "dispatches": len(c),
               "avg_us": sum(c) / len(c), "min_us": min(c), "max_us": max(c), "total_us": sum(c)}
        m = re.search(r"k_conv_wsINS_3GeoILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEELb(\d)ELb(\d)(?:ELi(\d))?", key[0])
        if m:
            S, R, ci, co, ipi, waves, tps, res, f32o = map(int, m.groups()[:9])
            mode = int(m.group(10) or 0)                                 # 0: gradient-free chain, 1: training forward (+ z), 2: input gradient
            row["shape"] = (("ws", "ws train", "ws dgrad", "ws +pool", "ws train +pool")[mode] + f" {ci}->{co} {S}x{S}" + (" +res" if res else "") + (" +f32" if f32o else ""))
            if a.batch:
                row["flops"] = 2.0 * a.batch * S * S * 9 * ci * co
                row["tflops_f32_equivalent"] = row["flops"] / row["avg_us"] / 1e6
        m = re.search(r"k_wgrad_px<.*WCfg<(\d+), (\d+), (\d+), (\d+), (\d+)", key[0]) or re.search(r"k_wgrad_pxINS_4WCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", key[0])
        if m:
            S, NI, R, co, ci = map(int, m.groups())
            row["shape"] = f"wgrad px {ci}->{co} {S}x{S}"
            if a.batch:
                row["flops"] = 2.0 * a.batch * S * S * 9 * ci * co
                row["tflops_f32_equivalent"] = row["flops"] / row["avg_us"] / 1e6
        out.append(row)
tot = sum(r["total_us"] for r in out)
out.sort(key=lambda r: -r["total_us"])
W_ = 66
print(f"{'kernel':{W_}s} {'grid':>9s} {'wg':>5s} {'n':>5s} {'avg us':>9s} {'min':>9s} {'max':>9s} {'share':>6s}  shape / TFLOP/s")
for r in out:
    extra = r.get("shape", "") + (f"  {r['tflops_f32_equivalent']:.0f} TF/s" if "tflops_f32_equivalent" in r else "")
    print(f"{r['kernel'][:W_]:{W_}s} {r['grid']:>9s} {r['workgroup']:>5s} {r['dispatches']:5d} {r['avg_us']:9.1f} {r['min_us']:9.1f} {r['max_us']:9.1f} {100 * r['total_us'] / tot:5.1f}%  {extra}")
if a.json:
    json.dump({"dir": a.dir, "batch": a.batch, "rows": out}, open(a.json, "w"), indent=1)

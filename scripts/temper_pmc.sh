#!/usr/bin/env bash
# HBM bytes per step of the sliding modes' rollout kernel (k_obs_roll_slide), as scripts/profile_r04.sh does for k_obs_roll: separate
# --pmc FETCH_SIZE / WRITE_SIZE / kernel-trace runs of a program that launches only that kernel.  Writes gpurun_out/r04/temper_pmc/.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/temper_pmc
mkdir -p "$O"
B="python3 bench.py --only-rollout --mode temper --no-cpu-baseline --no-dqn --steps 64 --warmup 64 --repeats 3 --settle-steps 1280"
rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- $B > "$O/trace.log" 2>&1 || echo trace failed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- $B > "$O/fetch.log" 2>&1 || echo fetch failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- $B > "$O/write.log" 2>&1 || echo write failed
python3 - <<'PY'
import csv, glob
def counter(d, name):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "k_obs_roll_slide" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]
def trace(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_obs_roll_slide" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
O = "gpurun_out/r04/temper_pmc"
F, W, T = counter(O + "/fetch", "FETCH_SIZE")[1:], counter(O + "/write", "WRITE_SIZE")[1:], trace(O + "/trace")[1:]
n = min(len(F), len(W), len(T))
fb, wb, us = sum(F[:n]) / n * 2048.0, sum(W[:n]) / n * 1024.0, sum(T[:n]) / n
alg = 2060 * 65536 * 64
print(f"k_obs_roll_slide, temper, 65 536 x 24x24, 64 steps per launch, {n} dispatches: {us:.1f} us per launch; read {fb / 64e6:.1f} MB + written {wb / 64e6:.1f} MB "
      f"= {(fb + wb) / 64e6:.1f} MB per step = {(fb + wb) / alg:.2f} x the 135.0 MB of the byte model; {(fb + wb) / us / 1e3:.0f} GB/s of counted traffic "
      f"({(fb + wb) / us / 1e3 / 8000:.3f} of 8 TB/s), {alg / us / 1e3:.0f} GB/s of algorithmic bytes ({alg / us / 1e3 / 8000:.3f})")
PY
find "$O" -name "*.db" -delete

#!/usr/bin/env bash
# Every bench.py line profiles/README.md lists for round 4, in one gpurun call (from the repo root).  Writes gpurun_out/r04/lines/.
set -uo pipefail
O=gpurun_out/r04/lines
mkdir -p "$O"
run() { local name=$1; shift; python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err" || echo "$name failed"; tail -c 300 "$O/$name.err" | grep -i "error\|traceback" || true; }
run r04_bench
run r04_bench_driver_invocation --steps 20 --warmup 5
run r04_bench_nonreversing --no-dqn --no-cpu-baseline --actions nonreversing
run r04_bench_planes4 --no-dqn --no-cpu-baseline --obs planes4
run r04_bench_config2_env_only --no-dqn --no-cpu-baseline --envs 4096 --width 10
run r04_bench_config5_env_only --no-dqn --no-cpu-baseline --envs 16384 --width 32
run r04_bench_temper --no-dqn --no-cpu-baseline --mode temper
run r04_bench_ice --no-dqn --no-cpu-baseline --mode ice
run r04_bench_1000_launches --no-dqn --no-cpu-baseline --steps 64000 --warmup 6400 --repeats 5
run r04_bench_config1_facade --config1 --steps 300
run r04_bench_acktr_config5 --acktr
python scripts/acktr_update_breakdown.py > "$O/r04_acktr_update_breakdown.txt" 2>&1 || echo "breakdown failed"
python scripts/gram_bench.py > "$O/r04_kfac_gram_bench.txt" 2>&1 || echo "gram bench failed"
python scripts/learn_cpu_profile.py 10 4096 > "$O/r04_learn_host_side_10x10.txt" 2>&1 || echo "host profile failed"
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04/lines/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = d.get("roofline") or {}
    extra = ""
    if "cold_start" in d: extra += f" cold {d['cold_start']['roofline']['frac']:.3f}"
    if "temper" in d and isinstance(d["temper"], dict) and "roofline" in d["temper"]: extra += f" temper {d['temper']['roofline']['frac']:.3f}"
    for k in ("dqn", "dqn_config3"):
        if k in d and isinstance(d[k], dict) and "learner_saturated" in d[k]:
            ls = d[k]["learner_saturated"]
            extra += f" | {k}: {d[k]['value']:.3g} tr/s loop {d[k]['roofline']['frac']:.3f} policy {d[k]['policy_rollout']['value']:.3g} ({d[k]['policy_rollout'].get('frac_of_f16x3_peak', 0):.3f}) learner {ls['ms_per_learn_step']:.2f} ms {ls['roofline']['frac']:.3f}"
    print(f"{os.path.basename(f):40s} value {d.get('value', 0):.4g} frac {r.get('frac', float('nan')) if r else float('nan'):.3f}{extra}")
PY

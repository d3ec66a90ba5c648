#!/usr/bin/env bash
# Round-4 rocprofv3 evidence (run through gpurun from the repo root).  Part A: the env rollout (kernel trace + PMC passes of a
# program that launches ONLY the timed kernel; both variants, the driver's 20 steps per launch and the default 64).  Part B: the
# learner and the policy forward per kernel.  Writes under gpurun_out/r04/prof; the summaries are copied to profiles/ by hand.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/prof
mkdir -p "$O"
PART=${1:-AB}
if [[ $PART == *A* ]]; then
  recs=()
  for cfg in "plain 20 5" "plain 64 64" "resident 20 5"; do
    set -- $cfg
    B="python3 bench.py --only-rollout --rollout-variant $1 --no-cpu-baseline --no-dqn --steps $(( $2 * 5 )) --warmup $3 --repeats 2"
    [[ $2 == 20 ]] && B="python3 bench.py --only-rollout --rollout-variant $1 --no-cpu-baseline --no-dqn --steps 20 --warmup $3 --repeats 6"
    t=$1_$2
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$t" -- $B > "$O/trace_$t.log" 2>&1 || echo "trace_$t failed"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$t" -- $B > "$O/fetch_$t.log" 2>&1 || echo "fetch_$t failed"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_$t" -- $B > "$O/write_$t.log" 2>&1 || echo "write_$t failed"
    recs+=(--record $1 $2 1 "$O/fetch_$t" "$O/write_$t" "$O/trace_$t")
  done
  python3 scripts/rollout_pmc.py "$O/r04_rollout_pmc.json" 4 "${recs[@]}" > "$O/rollout_pmc.txt" 2>&1 || echo "rollout_pmc failed"
  python3 scripts/kstats.py "$O/trace_plain_20" --filter k_obs > "$O/r04_kernel_trace_rollout_20.txt" 2>&1
  cp $(ls "$O"/trace_plain_20/*/*kernel_stats.csv | head -1) "$O/r04_kernel_stats_rollout_20.csv" 2>/dev/null
  # the driver's own invocation (everything bench.py launches), kernel trace + stats
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_driver" -- python3 bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 > "$O/trace_driver.log" 2>&1 || echo "trace_driver failed"
  python3 scripts/kstats.py "$O/trace_driver" --filter k_ > "$O/r04_kernel_trace_driver_invocation.txt" 2>&1
  cp $(ls "$O"/trace_driver/*/*kernel_stats.csv | head -1) "$O/r04_kernel_stats_driver_invocation.csv" 2>/dev/null
fi
if [[ $PART == *B* ]]; then
  for W in 10 24; do
    rocprofv3 --kernel-trace --output-format csv -d "$O/learn$W" -- python3 scripts/learn_prof.py $W 4096 8 > "$O/learn$W.log" 2>&1 || echo "learn$W failed"
    python3 scripts/kstats.py "$O/learn$W" --batch 4096 --skip-first 3 > "$O/r04_learn_${W}x${W}_kernel_rows.txt" 2>&1
  done
  for cfg in "8192 12" "2048 26"; do
    set -- $cfg
    rocprofv3 --kernel-trace --output-format csv -d "$O/infer$2" -- python3 scripts/infer_prof.py $1 $2 20 > "$O/infer$2.log" 2>&1 || echo "infer$2 failed"
    python3 scripts/kstats.py "$O/infer$2" --batch $1 --skip-first 2 > "$O/r04_infer_${2}x${2}_kernel_rows.txt" 2>&1
  done
  python3 scripts/trunk_px_bench.py 4096 12 > "$O/r04_trunk_px_bench_12.txt" 2>&1
  python3 scripts/trunk_px_bench.py 4096 26 > "$O/r04_trunk_px_bench_26.txt" 2>&1
fi
find "$O" -name "*.csv" -size +20M -delete
find "$O" -name "*.db" -delete
ls "$O"

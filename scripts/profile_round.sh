#!/usr/bin/env bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench, separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the rollout at the
#   default (64 steps per launch) and the driver's (20) invocation, and of the DQN loop's kernels.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02}/prof
mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --no-cpu-baseline > "$O/stats.log" 2>&1 || echo "stats failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats20" -- python3 bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 > "$O/stats20.log" 2>&1 || echo "stats20 failed"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$O/pmc64_$c" -- python3 bench.py --no-cpu-baseline --no-dqn --steps 128 --warmup 64 --repeats 2 > "$O/pmc64_$c.log" 2>&1 || echo "pmc64 $c failed"
  rocprofv3 --pmc $c --output-format csv -d "$O/pmc20_$c" -- python3 bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 --repeats 2 > "$O/pmc20_$c.log" 2>&1 || echo "pmc20 $c failed"
  rocprofv3 --pmc $c --output-format csv -d "$O/pmcdqn_$c" -- python3 bench.py --dqn --steps 6 --repeats 1 > "$O/pmcdqn_$c.log" 2>&1 || echo "pmcdqn $c failed"
done
find "$O" -name "*.csv" -size +30M -delete
ls -la "$O"/*/*/ 2>/dev/null | head -40

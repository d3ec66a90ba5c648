#!/usr/bin/env bash
# Round-3 rocprofv3 evidence (run through gpurun from the repo root): per-dispatch kernel rows and HBM bytes.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/prof
mkdir -p "$O"
B="python3 bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- $B > "$O/trace20.log" 2>&1 || echo "trace20 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch20" -- $B > "$O/fetch20.log" 2>&1 || echo "fetch20 failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write20" -- $B > "$O/write20.log" 2>&1 || echo "write20 failed"
python3 scripts/pmc_dispatch.py "$O/fetch20" "$O/write20" "$O/trace20" "$O/r03_rollout_hbm_per_dispatch_20.json" k_obs > "$O/pmc20.txt" 2>&1
python3 scripts/kstats.py "$O/trace20" --filter k_obs > "$O/r03_kernel_trace_driver_invocation.txt" 2>&1
cp $(ls "$O"/trace20/*/*kernel_stats.csv | head -1) "$O/r03_kernel_stats_driver_invocation.csv" 2>/dev/null
# Net.infer per layer (B fixed: FLOPs per row)
for cfg in "8192 12" "2048 26"; do
  set -- $cfg
  rocprofv3 --kernel-trace --output-format csv -d "$O/infer$2" -- python3 scripts/infer_prof.py $1 $2 20 > "$O/infer$2.log" 2>&1 || echo "infer$2 failed"
  python3 scripts/kstats.py "$O/infer$2" --batch $1 --skip-first 2 --json "$O/r03_infer_${2}x${2}_kernel_rows.json" > "$O/r03_infer_${2}x${2}_kernel_rows.txt" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/inferF$2" -- python3 scripts/infer_prof.py $1 $2 6 > /dev/null 2>&1 || echo "inferF failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/inferW$2" -- python3 scripts/infer_prof.py $1 $2 6 > /dev/null 2>&1 || echo "inferW failed"
  rocprofv3 --kernel-trace --output-format csv -d "$O/inferT$2" -- python3 scripts/infer_prof.py $1 $2 6 > /dev/null 2>&1 || echo "inferT failed"
  python3 scripts/pmc_dispatch.py "$O/inferF$2" "$O/inferW$2" "$O/inferT$2" "$O/r03_infer_${2}x${2}_hbm_per_dispatch.json" k_conv > /dev/null 2>&1
done
# the learner (config 2 and config 3 batch)
for W in 10 24; do
  rocprofv3 --kernel-trace --output-format csv -d "$O/learn$W" -- python3 scripts/learn_prof.py $W 4096 8 > "$O/learn$W.log" 2>&1 || echo "learn$W failed"
  python3 scripts/kstats.py "$O/learn$W" --batch 4096 --skip-first 3 > "$O/r03_learn_${W}x${W}_kernel_rows.txt" 2>&1
done
# ACKTR at BASELINE config 5 (16 384 envs x 32x32): one iteration's kernels
rocprofv3 --kernel-trace --output-format csv -d "$O/acktr" -- python3 bench.py --acktr --acktr-iterations 1 > "$O/acktr.log" 2>&1 || echo "acktr failed"
python3 scripts/kstats.py "$O/acktr" > "$O/r03_acktr_config5_kernel_rows.txt" 2>&1
python3 scripts/ws_layer_bench.py 8192 12 > "$O/r03_ws_layer_bench_12.txt" 2>&1
python3 scripts/ws_layer_bench.py 2048 26 > "$O/r03_ws_layer_bench_26.txt" 2>&1
find "$O" -name "*.csv" -size +20M -delete
find "$O" -name "*.db" -delete
ls "$O"

#!/usr/bin/env bash
# Same-box A/B of rollout-kernel builds (run through gpurun from the repo root): the in-tree libtron_hip.so against diagnostic
# builds in scratch_bin/ selected with TRON_HIP_LIB — e.g. scratch_bin/libtron_base.so = csrc/tron_env.hip of the previous commit
# compiled and linked against build/*.o (as scripts/ws_ablate.sh does for the conv kernels).  Prints value / roofline.frac /
# kernel_ms of `bench.py --no-dqn --no-cpu-baseline` at the driver's 20-step invocation (three rounds, interleaved), at the
# default 320 steps, and in temper mode.  Used for the kernel-argument re-read of k_obs_roll.
run() { python3 bench.py --steps $2 --warmup $3 --no-dqn --no-cpu-baseline $4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', $2, '$4', d['value'], r['frac'], r['kernel_ms'])"; }
BASE=${BASE:-scratch_bin/libtron_base.so}
for i in 1 2 3; do TRON_HIP_LIB=$BASE run base 20 5; run tree 20 5; done
for i in 1 2; do TRON_HIP_LIB=$BASE run base 320 64; run tree 320 64; done
for i in 1 2; do TRON_HIP_LIB=$BASE run base 320 64 "--mode temper"; run tree 320 64 "--mode temper"; done

#!/usr/bin/env bash
# Builds libtron_hip.so (gfx950 only) in-tree so it travels with the snapshot.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
    -o libtron_hip.so tron_env.hip tron_replay.hip tron_minimax.hip tron_kfac.hip tron_nn.hip tron_conv.hip tron_conv_f16.hip tron_conv_wgrad.hip tron_head.hip "$@"
echo "built $(pwd)/libtron_hip.so"

#!/usr/bin/env bash
# Diagnostic builds of csrc/tron_conv_ws.hip (TRON_WS_ABLATE = what is left out; results are wrong on purpose) linked
# against the regular objects, into scratch_bin/libtron_ws_ablate<N>.so.  Run on the build host; then on the GPU box:
#   for n in 0 1 2 3 4 5 6; do TRON_HIP_LIB=scratch_bin/libtron_ws_ablate$n.so python scripts/ws_layer_bench.py 8192 12 conv5; done
set -euo pipefail
cd "$(dirname "$0")/../deep-q-learning_tron_amd/csrc"
mkdir -p ../../scratch_bin
objs=$(ls build/*.o | grep -v tron_conv_ws.o)
if [ "${1:-}" = flags ]; then      # ws_ablate.sh flags <name> <-D...>: any other diagnostic build
  name=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c tron_conv_ws.hip -o ../../scratch_bin/ws_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../scratch_bin/libtron_ws_$name.so $objs ../../scratch_bin/ws_$name.o && rm ../../scratch_bin/ws_$name.o
  exit 0
fi
if [ "${1:-}" = poolstamps ]; then   # the WS_POOL kernel (tron_conv_ws_pool.hip) with stamps: scripts/ws_stamps.py 8192 12 conv6pool
  objs=$(ls build/*.o | grep -v tron_conv_ws_pool.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DTRON_WS_STAMPS -c tron_conv_ws_pool.hip -o ../../scratch_bin/ws_poolstamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../scratch_bin/libtron_ws_poolstamps.so $objs ../../scratch_bin/ws_poolstamps.o && rm ../../scratch_bin/ws_poolstamps.o
  exit 0
fi
if [ "${1:-}" = poolflags ]; then    # ws_ablate.sh poolflags <name> <-D...>: a diagnostic build of tron_conv_ws_pool.hip
  name=$2; shift 2
  objs=$(ls build/*.o | grep -v tron_conv_ws_pool.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c tron_conv_ws_pool.hip -o ../../scratch_bin/wsp_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../scratch_bin/libtron_wsp_$name.so $objs ../../scratch_bin/wsp_$name.o && rm ../../scratch_bin/wsp_$name.o
  exit 0
fi
if [ "${1:-}" = stamps ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DTRON_WS_STAMPS -c tron_conv_ws.hip -o ../../scratch_bin/ws_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../scratch_bin/libtron_ws_stamps.so $objs ../../scratch_bin/ws_stamps.o && rm ../../scratch_bin/ws_stamps.o
  exit 0
fi
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DTRON_WS_ABLATE=$n -c tron_conv_ws.hip -o ../../scratch_bin/ws_ablate$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../scratch_bin/libtron_ws_ablate$n.so $objs ../../scratch_bin/ws_ablate$n.o && rm ../../scratch_bin/ws_ablate$n.o ) &
done
wait
ls -la ../../scratch_bin/libtron_ws_ablate*.so

#!/usr/bin/env bash
# Builds libtron_hip.so (gfx950 only) in-tree so it travels with the snapshot.  One object per source, compiled in
# parallel and only when the source (or a header) is newer; extra arguments (-D...) force a full rebuild with them.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
SRCS="tron_env.hip tron_replay.hip tron_minimax.hip tron_kfac.hip tron_kfac_px.hip tron_nn.hip tron_conv.hip tron_conv_f16.hip tron_conv_ws.hip tron_conv_ws_pool.hip tron_conv_ws_train.hip tron_conv_wgrad.hip tron_conv_wgrad_rows.hip tron_head.hip tron_dqn.hip"
OBJDIR=build
mkdir -p "$OBJDIR"
STAMP="$OBJDIR/.flags"
if [ "$(cat "$STAMP" 2>/dev/null)" != "$FLAGS $*" ]; then rm -f "$OBJDIR"/*.o; echo "$FLAGS $*" > "$STAMP"; fi
newest_hdr=$(ls -t *.hpp ../../include/*.h | head -1)
pids=()
for s in $SRCS; do
    o="$OBJDIR/${s%.hip}.o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
        "$HIPCC" $FLAGS -c "$s" -o "$o" "$@" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=""
for s in $SRCS; do objs="$objs $OBJDIR/${s%.hip}.o"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -o libtron_hip.so $objs
echo "built $(pwd)/libtron_hip.so"

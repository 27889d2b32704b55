#!/bin/bash
# Builds libmcgen_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable ${MCGEN_EXTRA_FLAGS:-}"
mkdir -p build
pids=()
for f in conv_fused conv_skinny conv_smap conv_px1 conv_c8 conv_head wgrad wgrad_multi wgrad_c8 small_ops glow_ops pixelcnn_ops; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ mcgen_common.h -nt build/$f.o ] || [ conv_tile.h -nt build/$f.o ] || [ ../../include/mcgen_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmcgen_hip.so build/conv_fused.o build/conv_skinny.o build/conv_smap.o build/conv_px1.o build/conv_c8.o build/conv_head.o build/wgrad.o build/wgrad_multi.o build/wgrad_c8.o build/small_ops.o build/glow_ops.o build/pixelcnn_ops.o
echo "built $(pwd)/libmcgen_hip.so"

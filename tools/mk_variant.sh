#!/bin/bash
# Build container: a variant of libmcgen_hip.so with ONE source recompiled under extra flags -> csrc/build/<name>.so
# usage: tools/mk_variant.sh <name> <source stem, e.g. wgrad_multi> "<extra hipcc flags>"
set -euo pipefail
cd "$(dirname "$0")/../multimodal-controller-for-generative-models_amd/csrc"
NAME=$1; SRC=$2; EXTRA=${3:-}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $EXTRA -c $SRC.hip -o build/$SRC.$NAME.o
OBJS=""
for f in conv_fused conv_skinny conv_smap conv_px1 conv_c8 conv_head wgrad wgrad_multi wgrad_c8 small_ops glow_ops pixelcnn_ops; do
  if [ $f = $SRC ]; then OBJS="$OBJS build/$SRC.$NAME.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$NAME.so $OBJS
echo "built build/$NAME.so"

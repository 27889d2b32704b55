#!/bin/bash
# build the HIP library here (cross-compile), then run a command on the GPU box
set -e
cd "$(dirname "$0")/.."
multimodal-controller-for-generative-models_amd/csrc/build.sh >/dev/null
T=${GPU_TIMEOUT:-900}
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"

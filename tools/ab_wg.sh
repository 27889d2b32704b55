#!/bin/bash
# Runs ON THE GPU BOX: tools/bench_wgmulti.py for every build of the library under csrc/build/ab_*.so (made in the build
# container), interleaved over `rounds`.  usage: tools/ab_wg.sh [rounds] [reps]
D=multimodal-controller-for-generative-models_amd/csrc
R=${1:-2}; REPS=${2:-20}
cp $D/libmcgen_hip.so $D/build/_keep.so
for i in $(seq 1 $R); do
  for so in $D/build/ab_*.so; do
    cp $so $D/libmcgen_hip.so
    echo "== round $i  $(basename $so)"
    python tools/bench_wgmulti.py $REPS 2>/dev/null
  done
done
cp $D/build/_keep.so $D/libmcgen_hip.so

#!/bin/bash
# Runs ON THE GPU BOX: per-shape table rows matching a pattern for each library build given (csrc/build/<name>.so)
# usage: tools/so_shapes.sh <grep pattern> <name> [<name> ...]
D=$GRAFT_REPO_ROOT/multimodal-controller-for-generative-models_amd/csrc
PAT=$1; shift
cp $D/libmcgen_hip.so /tmp/keep.so
for v in "$@"; do
  cp $D/build/$v.so $D/libmcgen_hip.so
  echo "== $v"
  MCGEN_PROF_SHAPES=1 MCGEN_TUNING=1 python $GRAFT_REPO_ROOT/tools/shape_table.py 2>/dev/null | grep -E "$PAT"
done
cp /tmp/keep.so $D/libmcgen_hip.so

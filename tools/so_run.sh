#!/bin/bash
# Runs ON THE GPU BOX: a command once per library build given (csrc/build/<name>.so swapped in), the shipped library restored.
# usage: tools/so_run.sh "<command>" <name> [<name> ...]      ('base' = the shipped library)
D=$GRAFT_REPO_ROOT/multimodal-controller-for-generative-models_amd/csrc
CMD=$1; shift
cp $D/libmcgen_hip.so /tmp/keep.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/keep.so $D/libmcgen_hip.so; else cp $D/build/$v.so $D/libmcgen_hip.so; fi
  echo "== $v"
  bash -c "$CMD" 2>&1 | tail -8
done
cp /tmp/keep.so $D/libmcgen_hip.so

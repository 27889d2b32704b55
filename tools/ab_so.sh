#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of two builds of libmcgen_hip.so (csrc/build/prev.so vs csrc/build/cur.so, both made in
# the build container) on the headline bench.  usage: tools/ab_so.sh [rounds] [extra bench args]
D=multimodal-controller-for-generative-models_amd/csrc
R=${1:-3}; shift 1 || true
for i in $(seq 1 $R); do
  for v in prev cur; do
    cp $D/build/$v.so $D/libmcgen_hip.so
    ms=$(python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f%s' % (d['ms_per_step'], '' if d['config'].get('graph_replay', True) else ' (EAGER: capture failed)'))")
    echo "round $i  [$v]  $ms ms/step"
  done
done
cp $D/build/cur.so $D/libmcgen_hip.so

#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B rounds of the headline bench in one call (devices differ by up to ~10 %, so only
# same-box, interleaved comparisons mean anything).  usage: tools/ab_bench.sh "ENV_A" "ENV_B" [rounds] [extra bench args]
A="$1"; B="$2"; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for cfg in "$A" "$B"; do
    ms=$(env $cfg python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f%s' % (d['ms_per_step'], '' if d['config'].get('graph_replay', True) else ' (EAGER: capture failed)'))")
    echo "round $i  [$cfg]  $ms ms/step"
  done
done

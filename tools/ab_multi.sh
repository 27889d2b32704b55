#!/bin/bash
# Runs ON THE GPU BOX: like tools/ab_bench.sh for ANY number of environment settings.
# usage: tools/ab_multi.sh <rounds> "ENV_A" "ENV_B" ["ENV_C" ...]   (MCGEN_* switches need MCGEN_TUNING=1)
R=$1; shift
for i in $(seq 1 $R); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 40 --warmup 5 ${AB_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f%s' % (d['ms_per_step'], '' if d['config'].get('graph_replay', True) else ' (EAGER: capture failed)'))")
    echo "round $i  [$cfg]  $ms ms/step"
  done
done

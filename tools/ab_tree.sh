#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of two TREES on the headline bench -- ab_prev/ (an export of an earlier commit with
# its own built library: git archive <rev> multimodal-controller-for-generative-models_amd mcgen_amd bench.py oracle include tests/golden_util.py profiles/traffic.json
# | tar -x -C ab_prev; git-ignored) against the working tree.  For changes that move the C ABI, where swapping the .so
# under one Python tree (tools/ab_so.sh) cannot work.  usage: tools/ab_tree.sh [rounds] [extra bench args]
R=${1:-3}; shift 1 || true
for i in $(seq 1 $R); do
  for v in ab_prev .; do
    ms=$(cd $GRAFT_REPO_ROOT/$v && python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 40 --warmup 5 "$@" 2>$GRAFT_REPO_ROOT/gpurun_out/ab_tree_err_$(basename $v).log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f%s' % (d['ms_per_step'], '' if d['config'].get('graph_replay', True) else ' (EAGER: capture failed)'))")
    echo "round $i  [$v]  $ms ms/step"
  done
done

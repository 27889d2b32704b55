#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of a short bench run; prints the conv / wgrad rows (tools/kstat.sh <tag>)
set -e
TAG=${1:-k}
TREE=$GRAFT_REPO_ROOT/${KSTAT_TREE:-.}            # (KSTAT_TREE=ab_prev: the exported earlier tree of tools/ab_tree.sh)
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $TREE/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --sustain-steps 0 --real-data uniform --pool 1 ${KSTAT_ARGS:-} > $OUT/bench.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:86].ljust(86), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), r['Percentage'])
PY

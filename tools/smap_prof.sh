#!/bin/bash
# Runs ON THE GPU BOX: parity test of conv_smap, then its kernel durations under rocprofv3 (tools/smap_prof.sh <tag>)
set -e
TAG=${1:-smap}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
python -m pytest $GRAFT_REPO_ROOT/tests/test_kernels_gpu.py -x -q -k "whole_image or skinny" 2>&1 | tail -n 5
python $GRAFT_REPO_ROOT/tools/bench_smap.py 200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_smap.py 100 > $OUT/bench.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us')
PY

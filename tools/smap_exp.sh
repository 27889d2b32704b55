#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel durations of tools/bench_smap.py for each library build given (csrc/build/<name>.so)
D=$GRAFT_REPO_ROOT/multimodal-controller-for-generative-models_amd/csrc
cp $D/libmcgen_hip.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $D/build/$v.so $D/libmcgen_hip.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/smapx_$v; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_smap.py 100 > $OUT/bench.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:2]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us')
PY
done
cp /tmp/keep.so $D/libmcgen_hip.so

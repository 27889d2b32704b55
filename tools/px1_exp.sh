#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel durations of tools/bench_px1.py for each library build given (csrc/build/<name>.so)
D=$GRAFT_REPO_ROOT/multimodal-controller-for-generative-models_amd/csrc
cp $D/libmcgen_hip.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $D/build/$v.so $D/libmcgen_hip.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/px1x_$v; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_px1.py 50 > $OUT/bench.log 2>&1
  f=$(find $OUT -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_px1' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# six phases of 55 launches (5 warm-up + 50): 16 fwd, 16 bwd, 8 fwd, 8 bwd, 4 fwd, 4 bwd
names = ['16x16 fwd', '16x16 bwd', '8x8 fwd', '8x8 bwd', '4x4 fwd', '4x4 bwd']
for i, nm in enumerate(names):
    seg = rows[i * 55 + 5:(i + 1) * 55]
    if seg:
        d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in seg)
        print(f'{nm:10s} median {d[len(d) // 2]:7.2f} us')
PY
done
cp /tmp/keep.so $D/libmcgen_hip.so

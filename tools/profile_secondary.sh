#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace stats + the bench line of the secondary workloads (BASELINE configs[0], [3], [4]).
# Outputs under gpurun_out/prof_<TAG>_<workload>/ ; tools/profile_summary_secondary.py copies the summaries into profiles/.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for WL in mcglow mcpixelcnn mcvae; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$WL
  mkdir -p $OUT
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 10 --warmup 3 > $OUT/bench_trace.log 2>&1
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 20 --warmup 5 > $OUT/bench_clean.json 2> $OUT/bench_clean.err
  tail -c 300 $OUT/bench_clean.json; echo
done

#!/bin/bash
# Runs ON THE GPU BOX (through tools/gpu.sh): kernel-trace stats of the headline bench and two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate passes, MI355X_MICROARCH.md "HBM") over one eager iteration.
# Outputs land under gpurun_out/prof_$TAG; tools/profile_summary.py turns them into profiles/ files.
set -e
TAG=${1:-final}
WL=${2:-cifar10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --sustain-steps 0 > $OUT/bench_trace.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --sustain-steps 0 --real-data uniform --pool 1 > $OUT/bench_fetch.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --sustain-steps 0 --real-data uniform --pool 1 > $OUT/bench_write.log 2>&1
# steady-state graph replay (what the timed loop runs: no eager pool passes, no event brackets): profiles/<tag>_steady_kernel_stats.csv
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/steady -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --sustain-steps 0 --real-data uniform --pool 1 > $OUT/bench_steady.log 2>&1
find $OUT -name "*.csv" | head -20
tail -c 400 $OUT/bench_trace.log

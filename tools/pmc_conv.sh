#!/bin/bash
# PMC pass over the conv microbenchmark (run ON the GPU box): tools/pmc_conv.sh <cfg> <layer> <outdir-tag>
set -e
CFG=$1; LAYER=$2; TAG=$3; EXTRA=$4
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d ${OUT}_a -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --cfgs $CFG --only $LAYER $EXTRA > ${OUT}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d ${OUT}_b -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --cfgs $CFG --only $LAYER $EXTRA > ${OUT}_b.log 2>&1

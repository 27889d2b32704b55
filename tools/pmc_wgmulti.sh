#!/bin/bash
# PMC passes over the mcgen_wgrad_multi microbenchmark (run ON the GPU box): tools/pmc_wgmulti.sh <outdir-tag>
set -e
TAG=${1:-wgm}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d ${OUT}_a -- python3 $GRAFT_REPO_ROOT/tools/bench_wgmulti.py 4 > ${OUT}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d ${OUT}_b -- python3 $GRAFT_REPO_ROOT/tools/bench_wgmulti.py 4 > ${OUT}_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_f -- python3 $GRAFT_REPO_ROOT/tools/bench_wgmulti.py 4 > ${OUT}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_w -- python3 $GRAFT_REPO_ROOT/tools/bench_wgmulti.py 4 > ${OUT}_w.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py ${OUT}_a ${OUT}_b ${OUT}_f ${OUT}_w

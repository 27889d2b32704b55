#!/bin/bash
# Runs ON THE GPU BOX: parity tests of conv_px1, then an A/B of the MCGlow step against ab_prev/ (tools/ab_tree.sh)
set -e
python -m pytest $GRAFT_REPO_ROOT/tests/test_kernels_gpu.py -x -q -k "resident_tile" 2>&1 | tail -n 5
python -m pytest $GRAFT_REPO_ROOT/tests/test_mcglow_gpu.py -x -q 2>&1 | tail -n 3
bash $GRAFT_REPO_ROOT/tools/ab_tree.sh 2 --workload mcglow

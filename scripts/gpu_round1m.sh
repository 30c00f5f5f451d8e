#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $O/gpu_multirank_m.log 2>&1
rc=$?; echo "pytest multirank exit $rc"; tail -30 $O/gpu_multirank_m.log | cut -c1-300
exit $rc

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sparse_product" > $O/y_sparse.log 2>&1
rc=$?; echo "pytest sparse exit $rc"; tail -15 $O/y_sparse.log | cut -c1-200
exit $rc

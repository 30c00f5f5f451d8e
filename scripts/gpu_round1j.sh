#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "spmv" > $O/gpu_spmv_j.log 2>&1
rc=$?; echo "pytest spmv exit $rc"; tail -5 $O/gpu_spmv_j.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python scripts/kernel_bench.py 126 > $O/kernel_bench_j.log 2>&1
rc=$?; echo "kernel bench exit $rc"; tail -1 $O/kernel_bench_j.log
exit $rc

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/gpu_kernels.log 2>&1
echo "pytest kernels exit $?"; tail -3 $O/gpu_kernels.log
timeout -k 10 300 python scripts/kernel_bench.py 126 > $O/kernel_bench.log 2>&1
echo "kernel bench exit $?"; tail -2 $O/kernel_bench.log
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_b.log 2>&1
echo "bench exit $?"; tail -1 $O/bench_b.log
timeout -k 10 900 python -m pytest tests/test_gpu_geneo.py -m gpu -x -q > $O/gpu_geneo.log 2>&1
echo "pytest geneo exit $?"; tail -3 $O/gpu_geneo.log

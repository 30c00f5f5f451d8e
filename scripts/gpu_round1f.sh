#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests_f.log 2>&1
echo "pytest exit $?"; tail -3 $O/gpu_tests_f.log
timeout -k 10 600 python bench.py > $O/bench_f.log 2>&1
echo "bench exit $?"; tail -1 $O/bench_f.log
timeout -k 10 900 python bench.py --n-per-gpu 184 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_f_184.log 2>&1
echo "bench 184 exit $?"; tail -1 $O/bench_f_184.log
timeout -k 10 900 python bench.py --n-per-gpu 184 --steps 1 --warmup 0 --no-cpu-baseline --dls1-pc jacobi --els2-pc cheb > $O/bench_f_184_jacobi.log 2>&1
echo "bench 184 jacobi exit $?"; tail -1 $O/bench_f_184_jacobi.log

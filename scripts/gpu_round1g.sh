#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests_g.log 2>&1
echo "pytest exit $?"; tail -3 $O/gpu_tests_g.log
GENEO_DEBUG=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_g.log 2> $O/bench_g.err
echo "bench exit $?"; tail -1 $O/bench_g.log | cut -c 600-1500; grep "\[amg\] host" $O/bench_g.err | tail -4
GENEO_DEBUG=1 timeout -k 10 900 python bench.py --n-per-gpu 184 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_g_184.log 2> $O/bench_g_184.err
echo "bench 184 exit $?"; tail -1 $O/bench_g_184.log | cut -c 600-1500; grep "\[amg\]" $O/bench_g_184.err | tail -10

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_tests_e.log 2>&1
echo "pytest exit $?"; tail -12 $O/gpu_tests_e.log
GENEO_DEBUG=1 timeout -k 10 600 python bench.py > $O/bench_e.log 2> $O/bench_e.err
echo "bench exit $?"; tail -1 $O/bench_e.log; grep "\[amg\]" $O/bench_e.err | head -8
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --els2-pc cheb > $O/bench_e_cheb.log 2>&1
echo "bench els2 cheb exit $?"; tail -1 $O/bench_e_cheb.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1e -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_rocprof_e.log 2>&1
echo "rocprof exit $?"

#!/bin/bash
# round-1 evidence set: -m gpu suite, default bench (with CPU baseline), rocprofv3 kernel stats of the same command,
# kernel micro-benchmarks, 184^3 run
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/r_gpu_tests.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -3 $O/r_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/r_bench_default.log 2>&1
rc=$?; echo "bench exit $rc"; tail -1 $O/r_bench_default.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/kernel_bench.py 126 > $O/r_kernel_bench.log 2>&1
rc=$?; echo "kernel bench exit $rc"; tail -1 $O/r_kernel_bench.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --n-per-gpu 184 --steps 1 --warmup 1 --no-cpu-baseline > $O/r_bench_184.log 2>&1
rc=$?; echo "bench 184 exit $rc"; tail -1 $O/r_bench_184.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/r_prof.log 2>&1
rc=$?; echo "rocprof exit $rc"; tail -1 $O/r_prof.log | cut -c1-300
find $O/r_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r_kernel_stats.csv
exit $rc

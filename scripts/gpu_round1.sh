#!/bin/bash
# One gpurun call: GPU parity tests -> small bench sanity -> default bench -> rocprofv3 kernel trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
echo "pytest exit $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout -k 10 300 python bench.py --n-per-gpu 48 --steps 1 --warmup 0 --cpu-sample-n 16 > $O/bench_small.log 2>&1
echo "bench small exit $?"; tail -3 $O/bench_small.log
timeout -k 10 900 python bench.py > $O/bench_default.log 2>&1
echo "bench default exit $?"; tail -3 $O/bench_default.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_rocprof.log 2>&1
echo "rocprof exit $?"; tail -2 $O/bench_rocprof.log
ls -R $O/prof_r1 | head -20

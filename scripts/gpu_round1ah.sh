#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1000 python bench.py --n-per-gpu 232 --steps 1 --warmup 0 --no-cpu-baseline > $O/ah_bench_232.log 2>&1
rc=$?; echo "bench 232 exit $rc"; tail -1 $O/ah_bench_232.log | cut -c1-1500
rocm-smi --showmemuse 2>/dev/null | tail -5
exit $rc

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_p.log 2>&1
rc=$?; echo "bench exit $rc"; tail -3 $O/bench_p.log | cut -c1-2500
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_geneo.py -m gpu -x -q > $O/gpu_geneo_p.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -3 $O/gpu_geneo_p.log
exit $rc

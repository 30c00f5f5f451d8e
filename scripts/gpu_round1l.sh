#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_all_l.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -4 $O/gpu_all_l.log
[ $rc -eq 0 ] || exit $rc
GENEO_DEBUG=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_l_debug.log 2>&1
rc=$?; echo "bench debug exit $rc"; grep -E "^\[amg\]|iterations .* s: host|^\[lobpcg tau\] it (0|10|20|30|40|50|60) " $O/bench_l_debug.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_l.log 2>&1
rc=$?; echo "bench exit $rc"; tail -1 $O/bench_l.log
exit $rc

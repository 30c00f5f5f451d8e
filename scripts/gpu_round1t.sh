#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
GENEO_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_t.log 2>&1
rc=$?; echo "exit $rc"; grep -E "lobpcg tau\] it [0-9]*[05] " $O/bench_t.log | sed 's/maxres:.*| locked/locked/' | cut -c1-120 | head -20
grep -E "^\[amg\] A_Neu|iterations .* s: host" $O/bench_t.log
exit $rc

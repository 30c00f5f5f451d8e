#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
GENEO_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/z_bench.log 2>&1
rc=$?; echo "exit $rc"; grep -E "^\[setup\]|^\[amg" $O/z_bench.log | tail -14
exit $rc

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
GENEO_DEBUG=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --n-per-gpu 64 > $O/bench_q.log 2>&1
rc=$?; echo "bench exit $rc"; grep -E "^\[graph\]" $O/bench_q.log | head; tail -1 $O/bench_q.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print(j['roofline'])"
exit $rc

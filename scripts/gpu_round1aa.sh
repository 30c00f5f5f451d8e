#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gram or blockmul or block" > $O/aa_tests.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -3 $O/aa_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/kernel_bench.py 126 > $O/aa_kernel_bench.log 2>&1
rc=$?; echo "kernel bench exit $rc"; tail -1 $O/aa_kernel_bench.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print(j['gram96'], j['blockmul96x64'], j['spmm32'])"
exit $rc

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/ae_gpu_tests.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -3 $O/ae_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/ae_smoke.log 2>&1
rc=$?; echo "smoke exit $rc"; tail -2 $O/ae_smoke.log
exit $rc

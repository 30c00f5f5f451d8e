#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_geneo.py -m gpu -x -q -k "hub_row" > $O/ad_hub.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -12 $O/ad_hub.log | cut -c1-220
exit $rc

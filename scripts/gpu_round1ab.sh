#!/bin/bash
# the alternative code paths of the product (fallbacks) against the same parity suite
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for V in GENEO_AMG_HOST GENEO_NO_GRAPH GENEO_AMG_UNFUSED GENEO_E_COLUMNWISE GENEO_NO_PINNED_STAGING; do
  env $V=1 timeout -k 10 400 python -m pytest tests/test_gpu_geneo.py -m gpu -x -q > $O/ab_$V.log 2>&1
  rc=$?; echo "$V=1: exit $rc: $(tail -1 $O/ab_$V.log)"
  [ $rc -eq 0 ] || exit $rc
done

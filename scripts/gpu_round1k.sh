#!/bin/bash
# in-situ comparison of the sliced SpMV variants (all fine-level launches sampled: graphs off)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for V in 2 3 6 8; do
  GENEO_NO_GRAPH=1 GENEO_SELL_VARIANT=$V timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_k_v$V.log 2>&1
  rc=$?; echo "variant $V exit $rc"
  [ $rc -eq 0 ] || exit $rc
  python - <<PY
import json
j=json.loads(open("$O/bench_k_v$V.log").read().strip().splitlines()[-1])
r=j["roofline"]
print("variant $V: value %.0f GB/s avg %.4f ms (%d sampled) setup %.3f solve %.3f" % (j["value"], r["avg_launch_ms"], r["launches_timed"], j["setup_s"], j["solve_s"]))
PY
done

#!/bin/bash
# the coarse start on the 10 M-node graph workload (irregular rows, 1.25 M rows per subdomain): on (default) against off
set -o pipefail
O=gpurun_out/coarse_start_graph.log
: > $O
for cs in 0 750000; do
  echo "== graph 10M -geneo_eig_coarse_start $cs" | tee -a $O
  timeout -k 10 500 python bench.py --workload graph --steps 2 --warmup 1 --pc-args "-geneo_eig_coarse_start $cs" > gpurun_out/csg_$cs.json 2> gpurun_out/csg_$cs.err || exit 1
  python - gpurun_out/csg_$cs.json <<'PY' | tee -a $O
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "setup_breakdown_s")})
PY
done

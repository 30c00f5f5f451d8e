#!/bin/bash
# nesting bound of the coarse start on the whole 368^3 step: 4096 rows per subdomain (new default) against 20000
set -o pipefail
for mr in 20000 4096; do
  echo "== GENEO_COARSE_START_MIN_ROWS=$mr"
  GENEO_COARSE_START_MIN_ROWS=$mr timeout -k 10 500 python bench.py --steps 4 --warmup 2 > gpurun_out/nest_$mr.json 2> gpurun_out/nest_$mr.err || exit 1
  python - gpurun_out/nest_$mr.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "setup_breakdown_s")})
PY
done

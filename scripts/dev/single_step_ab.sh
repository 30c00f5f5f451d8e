#!/bin/bash
# single-step tail of the inner PCG (default from 4 M local rows) against pairs, whole 368^3 step and one rank of 8
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_geneo.py -m gpu -x -q -s -k "single_step" 2>&1 | tail -4 || exit 1
for args in "--steps 4 --warmup 2" "--one-rank-of 8 --steps 4 --warmup 2"; do
  for rows in 1000000000000 4000000; do
    echo "== bench.py $args  GENEO_DLS1_SINGLE_STEP_ROWS=$rows"
    GENEO_DLS1_SINGLE_STEP_ROWS=$rows timeout -k 10 500 python bench.py $args --no-cpu-baseline > gpurun_out/ss.json 2> gpurun_out/ss.err || { tail -5 gpurun_out/ss.err; exit 1; }
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/ss.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "local_solve_cg_iterations", "local_solves")})
PY
  done
done

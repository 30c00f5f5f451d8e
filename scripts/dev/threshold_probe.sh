#!/bin/bash
# where does the coarse start begin to pay?  8 subdomains of 74^3 (0.41 M rows) and 82^3 (0.55 M rows), off / on
set -o pipefail
for n in 144 160; do
  for cs in 0 1; do
    echo "== ${n}^3 in 8 subdomains, -geneo_eig_coarse_start $cs"
    timeout -k 10 400 python bench.py --scaling weak --n $n --steps 3 --warmup 1 --no-cpu-baseline --pc-args "-geneo_eig_coarse_start $cs" > gpurun_out/thr.json 2> gpurun_out/thr.err || { tail -5 gpurun_out/thr.err; exit 1; }
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/thr.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "setup_breakdown_s")})
PY
  done
done

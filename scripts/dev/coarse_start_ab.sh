#!/bin/bash
# A/B of the eigensolver's coarse start (-geneo_eig_coarse_start) on one rank's share of the 368^3 / 8 benchmark and on
# 8 subdomains of smaller grids (where does it stop paying?).
set -o pipefail
O=gpurun_out/coarse_start_ab.log
: > $O
one() {  # tag, threshold, env..., -- bench args
  tag=$1; cs=$2; shift 2
  echo "== $tag -geneo_eig_coarse_start $cs $ENVX" | tee -a $O
  env GENEO_DEBUG=1 $ENVX timeout -k 10 400 python bench.py "$@" --steps 2 --warmup 1 --pc-args "-geneo_eig_coarse_start $cs" > gpurun_out/cs_$tag.json 2> gpurun_out/cs_$tag.err || exit 1
  grep -E "coarse start|\[lobpcg.*iterations" gpurun_out/cs_$tag.err | tail -${LINES_:-4} | cut -c1-200 | tee -a $O
  python - gpurun_out/cs_$tag.json <<'PY' | tee -a $O
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "local_solve_cg_iterations", "setup_breakdown_s")})
PY
}
ENVX="" one r8_nested 1 --one-rank-of 8
ENVX="GENEO_COARSE_START_MIN_ROWS=100000000" one r8_single 1 --one-rank-of 8
ENVX="GENEO_COARSE_START_TOL=0.03" one r8_tol3e-2 1 --one-rank-of 8
ENVX="GENEO_COARSE_START_TOL=0.003" one r8_tol3e-3 1 --one-rank-of 8
ENVX="" one w126_off 0 --scaling weak
ENVX="" one w126_on 1 --scaling weak
ENVX="" one w184_off 0 --scaling weak --n 184
ENVX="" one w184_on 1 --scaling weak --n 184
ENVX="" one w232_off 0 --scaling weak --n 232
ENVX="" one w232_on 1 --scaling weak --n 232

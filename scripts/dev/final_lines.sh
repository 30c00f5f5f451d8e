#!/bin/bash
# the bench lines of the round's final source, one after the other (each a fresh process)
set -o pipefail
O=gpurun_out
run() { tag=$1; shift; echo "== $tag: bench.py $*"; timeout -k 10 500 python bench.py "$@" > $O/fin_$tag.json 2> $O/fin_$tag.err || exit 1
  python - $O/fin_$tag.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "device_mem_peak_gb")})
PY
}
run rank_of_8 --one-rank-of 8 --steps 5 --warmup 2
run weak_126 --scaling weak --steps 10 --warmup 3
run heat_126 --workload heat --steps 5 --warmup 2
run graph_10M --workload graph --steps 5 --warmup 2

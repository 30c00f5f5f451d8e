#!/bin/bash
# LOBPCG block width at 126^3 / 8 subdomains (gap-limited: 59 iterations with the 32-column block)
set -o pipefail
for extra in "" "-els2_eps_block 64"; do
  echo "== 126^3 weak, pc-args: '$extra'"
  timeout -k 10 400 python bench.py --scaling weak --steps 3 --warmup 1 --pc-args "$extra" > gpurun_out/blk.json 2> gpurun_out/blk.err || { tail -5 gpurun_out/blk.err; exit 1; }
  python - <<'PY'
import json
j = json.loads(open("gpurun_out/blk.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "local_solve_cg_iterations", "setup_breakdown_s")})
PY
done

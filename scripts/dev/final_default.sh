#!/bin/bash
# the driver's own invocation at N = 1, with the wall time of the whole command beside the JSON line
set -o pipefail
t0=$(date +%s.%N)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final368_bench.json 2> gpurun_out/final368_bench.err || exit 1
t1=$(date +%s.%N)
echo "bench.py --gpus 1 --steps 20 --warmup 5: wall $(echo "$t1 - $t0" | bc) s" | tee gpurun_out/driverlike_wall.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/final368_bench.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "device_mem_peak_gb")})
r = j["roofline"]
print("dominant", r["kernel"][:40], round(r["frac"], 3), "traffic", r.get("traffic"), [(k["kernel"][:14], round(k["frac"], 3), k.get("mfma_util")) for k in r["kernels"]])
PY

#!/bin/bash
# the driver's own invocation at N = 1, with the wall time of the whole command beside the JSON line
set -o pipefail
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final368_bench.json 2> gpurun_out/final368_bench.err || exit 1
python - <<'PY'
import json
j = json.loads(open("gpurun_out/final368_bench.json").read().strip().splitlines()[-1])
open("gpurun_out/driverlike_wall.txt", "w").write("bench.py --gpus 1 --steps 20 --warmup 5: wall %.0f s (bench_wall_s of the line: the whole process)\n" % j["bench_wall_s"])
print(open("gpurun_out/driverlike_wall.txt").read().strip())
print({k: j.get(k) for k in ("ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "eig_coarse_iterations", "local_solve_cg_iterations", "device_mem_peak_gb")})
r = j["roofline"]
print("dominant", r["kernel"][:40], round(r["frac"], 3), "traffic", r.get("traffic"), [(k["kernel"][:14], round(k["frac"], 3), k.get("mfma_util")) for k in r["kernels"]])
PY

#!/bin/bash
# inner-solver parameter sweep on one rank's share of the metric's configuration (bench.py --one-rank-of 8): solve time and
# inner iteration counts per variant, one line each to gpurun_out/tune_solve.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py --one-rank-of 8 --steps 2 --warmup 1 --pc-args "$1" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s setup %.4f solve %.4f inner its %d eig its %d levels %s' % (sys.argv[1] or '(default)', j['setup_s'], j['solve_s'], j['local_solve_cg_iterations'], j['eig_iterations'], j['amg_levels']))" "$1" | tee -a gpurun_out/tune_solve.log
}
for v in "$@"; do run "$v" || exit 1; done

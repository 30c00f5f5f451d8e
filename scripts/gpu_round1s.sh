#!/bin/bash
# inner tolerance of the local solves vs outer convergence; AMG host set-up after threading
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for T in 1e-10 1e-8 1e-6 1e-4; do
  GENEO_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --dls1-rtol $T > $O/bench_s_$T.log 2>&1
  rc=$?; echo "dls1-rtol $T exit $rc"
  [ $rc -eq 0 ] || exit $rc
  grep -E "A_Neu host" $O/bench_s_$T.log | tail -1
  tail -1 $O/bench_s_$T.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d inner %d value %.0f' % (j['setup_s'], j['solve_s'], j['iterations'], j['local_solve_cg_iterations'], j['value']), j['untimed_step_with_hip_graphs_s'])"
done

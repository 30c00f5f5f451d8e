#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
i=0
for A in "-dls1_check 4" "-dls1_check 2" "-dls1_check 6"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --pc-args "$A" > $O/aj_$i.log 2>&1
  rc=$?; echo "[$A] exit $rc"
  [ $rc -eq 0 ] || exit $rc
  tail -1 $O/aj_$i.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d inner %d' % (j['setup_s'], j['solve_s'], j['iterations'], j['local_solve_cg_iterations']), j['untimed_step_with_hip_graphs_s'])"
done

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
i=0
for A in "-amg_smooth_ratio 2.5" "-amg_smooth_ratio 4" "-amg_smooth_ratio 7" "-amg_smooth_ratio 12" "-amg_coarse_size 1200" "-amg_coarse_size 300"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --pc-args "$A" > $O/af_$i.log 2>&1
  rc=$?; echo "[$A] exit $rc"
  [ $rc -eq 0 ] || exit $rc
  tail -1 $O/af_$i.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d eig %d inner %d levels %d' % (j['setup_s'], j['solve_s'], j['iterations'], j['eig_iterations'], j['local_solve_cg_iterations'], j['amg_levels']), j['setup_breakdown_s']['eigensolve_lobpcg'])"
done

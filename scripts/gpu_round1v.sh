#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
i=0
for A in "" "-els2_amg_plain 1" "-els2_amg_plain 1 -dls1_amg_plain 1" "-dls1_amg_plain 1"; do
  i=$((i+1))
  GENEO_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --pc-args "$A" > $O/bench_v_$i.log 2>&1
  rc=$?; echo "[$A] exit $rc"
  [ $rc -eq 0 ] || exit $rc
  grep -E "A_Neu host|level-1 host" $O/bench_v_$i.log | tail -2
  tail -1 $O/bench_v_$i.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d eig %d inner %d value %.0f' % (j['setup_s'], j['solve_s'], j['iterations'], j['eig_iterations'], j['local_solve_cg_iterations'], j['value']), j['setup_breakdown_s'], j['untimed_step_with_hip_graphs_s'])"
done

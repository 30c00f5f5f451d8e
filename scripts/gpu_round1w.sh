#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/w_gpu_tests.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -2 $O/w_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for NPG in 126 184; do
GENEO_DEBUG=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --n-per-gpu $NPG > $O/w_bench_$NPG.log 2>&1
rc=$?; echo "bench $NPG exit $rc"; grep -E "^\[amg\] (A_Neu|level-1)" $O/w_bench_$NPG.log | tail -2
tail -1 $O/w_bench_$NPG.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d eig %d inner %d value %.0f' % (j['setup_s'], j['solve_s'], j['iterations'], j['eig_iterations'], j['local_solve_cg_iterations'], j['value']), j['setup_breakdown_s'], j['untimed_step_with_hip_graphs_s'])"
[ $rc -eq 0 ] || exit $rc
done

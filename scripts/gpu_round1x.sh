#!/bin/bash
# robustness at scale: GenEO-2 on the bench workload, heat (high contrast) and graph generators through the CLI driver
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --lvl SORAS,2 --tau 0.05 --pc-args "-geneo_gamma 1.05 -geneo_optim 0.5" > $O/x_bench_g2.log 2>&1
rc=$?; echo "bench GenEO-2 exit $rc"; tail -1 $O/x_bench_g2.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('  setup %.3f solve %.3f its %d eig %d dimE %d' % (j['setup_s'], j['solve_s'], j['iterations'], j['eig_iterations'], j['dimE']), j['converged'])" || tail -5 $O/x_bench_g2.log
timeout -k 10 500 python -m geneo4petsc_amd.driver --inpLibA "heat#--size#64#--dim#3#--kappa#100#minmax" --np 8 --parts 2,2,2 --metisNodal --addOverlap 2 --timing \
   -geneo_lvl ASM,1 -geneo_tau 0.1 -geneo_cut 20 -ksp_type cg > $O/x_driver_heat.log 2>&1
rc=$?; echo "driver heat exit $rc"; grep -E "^INFO|^TIME|Error" $O/x_driver_heat.log | cut -c1-220
timeout -k 10 500 python -m geneo4petsc_amd.driver --inpLibA "graph#--size#40000#--level#2#--noGround" --np 8 --metisNodal --addOverlap 1 --timing \
   -geneo_lvl RAS,1 -geneo_tau 0.2 -geneo_cut 10 > $O/x_driver_graph.log 2>&1
rc=$?; echo "driver graph exit $rc"; grep -E "^INFO|^TIME|Error" $O/x_driver_graph.log | cut -c1-220
exit 0

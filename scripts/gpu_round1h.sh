#!/bin/bash
# GPU check after GenEO-2 / driver: full -m gpu suite, a GenEO-2 driver run, the default bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_all_h.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -14 $O/gpu_all_h.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m geneo4petsc_amd.driver --inpLibA "laplacian#--size#48#--dim#3" --np 8 --parts 2,2,2 --metisNodal \
  --addOverlap 2 --timing -geneo_lvl SORAS,2 -geneo_tau 0.02 -geneo_gamma 1.05 -geneo_cut 20 -geneo_optim 0.5 -ksp_type cg > $O/driver_g2_h.log 2>&1
rc=$?; echo "driver exit $rc"; cat $O/driver_g2_h.log | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_h.log 2>&1
rc=$?; echo "bench exit $rc"; tail -1 $O/bench_h.log
exit $rc

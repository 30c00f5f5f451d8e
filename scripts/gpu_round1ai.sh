#!/bin/bash
# BASELINE config 4 in small: high-contrast heat problem, one-level vs two-level (GenEO) iteration counts
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for LVL in "ASM,0" "ASM,1"; do
  timeout -k 10 500 python -m geneo4petsc_amd.driver --inpLibA "heat#--size#80#--dim#3#--kappa#100#minmax" --np 8 --parts 2,2,2 --metisNodal --addOverlap 1 --timing \
     -geneo_lvl $LVL -geneo_tau 0.1 -geneo_cut 20 -ksp_type cg -ksp_rtol 1e-8 > $O/ai_heat_${LVL/,/_}.log 2>&1
  rc=$?; echo "$LVL exit $rc"; grep -E "^INFO: (geneo|setup|solve)|^TIME" $O/ai_heat_${LVL/,/_}.log | cut -c1-230
done
exit 0

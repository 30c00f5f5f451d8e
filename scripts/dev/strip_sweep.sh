#!/bin/bash
# Ceiling experiments of the sliced SpMM on one 6.4 M-row subdomain of the 368^3 decomposition (scripts/spmm_bench.py, ONE=1):
# the same launch on matrices that keep only some of the seven gathers, and small strips.
set -o pipefail
O=gpurun_out/strip_sweep2.log
: > $O
run() { echo "== $*" | tee -a $O; env ONE=1 REPS=10 "$@" timeout -k 10 200 python scripts/spmm_bench.py 368 2 32 2>> $O | tail -1 | tee -a $O || exit 1; }
run MATRIX=diag
run MATRIX=tri
run MATRIX=lines
run MATRIX=planes
run GENEO_SPMM_STRIP=768
run GENEO_SPMM_STRIP=1024 GENEO_SPMM_U=1
run GENEO_SPMM_STRIP=3072 GENEO_SPMM_U=4

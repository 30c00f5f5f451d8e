#!/bin/bash
# HBM traffic and L2 behaviour of the 32-column SpMM (separate PMC passes, --kernel-trace only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_spmm_a $O/pmc_spmm_b $O/pmc_spmm_c
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_spmm_a -- python3 $R/scripts/kernel_bench.py 126 > $O/am_a.log 2>&1
rc=$?; echo "pass FETCH exit $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_spmm_b -- python3 $R/scripts/kernel_bench.py 126 > $O/am_b.log 2>&1
rc=$?; echo "pass WRITE exit $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_spmm_c -- python3 $R/scripts/kernel_bench.py 126 > $O/am_c.log 2>&1
rc=$?; echo "pass TCC exit $rc"
python3 $R/scripts/pmc_mfma_report.py $O/pmc_spmm_a $O/pmc_spmm_b $O/pmc_spmm_c > $O/am_report.json 2>&1
cat $O/am_report.json

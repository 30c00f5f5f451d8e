#!/bin/bash
# PMC passes for the MFMA block kernels on the bench shapes (counters in their own runs, --kernel-trace only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_mfma_a $O/pmc_mfma_b
timeout -k 10 400 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pmc_mfma_a -- python3 $R/scripts/kernel_bench.py 126 > $O/al_a.log 2>&1
rc=$?; echo "pass MfmaUtil exit $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_b -- python3 $R/scripts/kernel_bench.py 126 > $O/al_b.log 2>&1
rc=$?; echo "pass LDS exit $rc"; [ $rc -eq 0 ] || exit $rc
python3 $R/scripts/pmc_mfma_report.py $O/pmc_mfma_a $O/pmc_mfma_b > $O/al_report.json 2>&1
cat $O/al_report.json

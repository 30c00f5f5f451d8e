#!/bin/bash
# HBM traffic of the final SpMV kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_fetch $O/pmc_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/pmc_spmv.py 126 > $O/pmc_fetch.log 2>&1
rc=$?; echo "pmc fetch exit $rc"; tail -1 $O/pmc_fetch.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/scripts/pmc_spmv.py 126 > $O/pmc_write.log 2>&1
rc=$?; echo "pmc write exit $rc"; tail -1 $O/pmc_write.log
[ $rc -eq 0 ] || exit $rc
python3 $R/scripts/pmc_report.py $O/pmc_fetch $O/pmc_write 235645999 40000000 > $O/pmc_report_u.json 2>&1
cat $O/pmc_report_u.json

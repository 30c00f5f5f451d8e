#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_tests_d.log 2>&1
echo "pytest exit $?"; tail -12 $O/gpu_tests_d.log
timeout -k 10 600 python bench.py > $O/bench_d.log 2>&1
echo "bench exit $?"; tail -1 $O/bench_d.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --dls1-pc jacobi --els2-pc cheb > $O/bench_d_jacobi.log 2>&1
echo "bench jacobi exit $?"; tail -1 $O/bench_d_jacobi.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_rocprof_d.log 2>&1
echo "rocprof exit $?"; tail -1 $O/bench_rocprof_d.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/pmc_spmv.py 126 > $O/pmc_fetch.log 2>&1
echo "pmc fetch exit $?"; tail -1 $O/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/scripts/pmc_spmv.py 126 > $O/pmc_write.log 2>&1
echo "pmc write exit $?"; tail -1 $O/pmc_write.log
python3 $R/scripts/pmc_report.py $O/pmc_fetch $O/pmc_write 235645999 40000000 > $O/pmc_report.json 2>&1
cat $O/pmc_report.json

#!/bin/bash
# fused multigrid epilogues: kernel parity, full parity suite, bench with / without the fused cycle, kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_all_i.log 2>&1
rc=$?; echo "pytest gpu exit $rc"; tail -9 $O/gpu_all_i.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_i_fused.log 2>&1
rc=$?; echo "bench fused exit $rc"; tail -1 $O/bench_i_fused.log
[ $rc -eq 0 ] || exit $rc
GENEO_AMG_UNFUSED=1 timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_i_unfused.log 2>&1
rc=$?; echo "bench unfused exit $rc"; tail -1 $O/bench_i_unfused.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_i.log 2>&1
rc=$?; echo "rocprof exit $rc"; tail -1 $O/prof_i.log
find $O/prof_i -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_i.csv
exit $rc

#!/bin/bash
# One parametrised GPU-box script (replaces the per-experiment scripts of round 1).  Run through gpurun:
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh <task> [args]'
# Every task writes its logs under gpurun_out/<tag>_* (tag = $TAG, default the task name) and never retries.
#   tests [pytest args]        pytest -m gpu (default: the whole GPU suite)
#   bench [bench.py args]      one bench.py run, JSON line to gpurun_out/<tag>_bench.json
#   debug [bench.py args]      the same with GENEO_DEBUG=1 (set-up / LOBPCG / AMG phase lines on stderr)
#   prof  [bench.py args]      rocprofv3 --kernel-trace --stats of bench.py, per-kernel CSV to gpurun_out/<tag>_stats.csv
#   trace <anchor> <occ> <cnt> [bench.py args]  kernel timeline window of bench.py (gaps between kernels)
#   spmm  "<wpx list>" [args]  scripts/spmm_bench.py for each GENEO_SPMM_WPX in the list
#   pmc   <counter> <cmd...>   rocprofv3 --pmc <counter> (own pass, kernel trace only) of a python command
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
task=$1; shift
TAG=${TAG:-$task}
case $task in
  tests)
    if [ $# -eq 0 ]; then set -- tests -m gpu -x -q; fi
    timeout -k 10 ${LIMIT:-1100} python -m pytest "$@" > $O/${TAG}.log 2>&1; rc=$?
    tail -25 $O/${TAG}.log; exit $rc ;;
  bench|debug)
    [ $task = debug ] && export GENEO_DEBUG=1
    timeout -k 10 ${LIMIT:-600} python bench.py "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; rc=$?
    grep -E "^\[lobpcg.*iterations|^\[setup\]|^\[amg" $O/${TAG}_bench.err | tail -30
    tail -3 $O/${TAG}_bench.err
    python - $O/${TAG}_bench.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keep = ["value", "ms_per_step", "setup_s", "solve_s", "iterations", "dimE", "eig_iterations", "local_solve_cg_iterations", "device_mem_peak_gb", "eig_groups", "host_prep_s", "scaling",
            "setup_breakdown_s", "solve_breakdown_s", "parity_sample"]
    print(json.dumps({k: j.get(k) for k in keep}))
    print("roofline", json.dumps(j.get("roofline")))
except Exception as e:
    print("no JSON line:", e)
PY
    exit $rc ;;
  prof)
    rm -rf /tmp/prof_$TAG
    (cd /tmp && timeout -k 10 ${LIMIT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python3 $R/bench.py "$@" > $O/${TAG}_prof.json 2> $O/${TAG}_prof.err); rc=$?
    f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $O/${TAG}_stats.csv && head -25 $O/${TAG}_stats.csv | cut -c1-200
    tail -2 $O/${TAG}_prof.err; exit $rc ;;
  trace)  # kernel timeline of a window: trace <anchor kernel> <occurrence> <count> [bench.py args]
    anchor=$1; occ=$2; cnt=$3; shift 3
    rm -rf /tmp/trace_$TAG
    (cd /tmp && timeout -k 10 ${LIMIT:-600} rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$TAG -o run -- python3 $R/bench.py "$@" > $O/${TAG}_trace.json 2> $O/${TAG}_trace.err); rc=$?
    f=$(find /tmp/trace_$TAG -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && [ -n "$GAPS" ] && python3 scripts/trace_gaps.py $f $GAPS > $O/${TAG}_gaps.txt
    [ -n "$f" ] && python3 scripts/trace_window.py $f "$anchor" $occ $cnt > $O/${TAG}_window.txt && tail -3 $O/${TAG}_window.txt; [ -n "$f" ] && [ -n "$ANCHORB" ] && python3 scripts/trace_window.py $f "$ANCHORB" 0 0 > $O/${TAG}_intervals.txt
    exit $rc ;;
  spmm)
    list=$1; shift
    for w in $list; do
      GENEO_SPMM_WPX=$w timeout -k 10 300 python scripts/spmm_bench.py "$@" 2>&1 | tail -1 | tee -a $O/${TAG}.log || exit 1
    done ;;
  pmc)
    ctr=$1; shift
    rm -rf /tmp/pmc_$TAG
    (cd /tmp && timeout -k 10 ${LIMIT:-600} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o run -- "$@" > $O/${TAG}_$ctr.log 2>&1); rc=$?
    f=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $O/${TAG}_$ctr.csv && wc -l $O/${TAG}_$ctr.csv
    tail -3 $O/${TAG}_$ctr.log; exit $rc ;;
  pmcm)   # MFMA utilisation of the Rayleigh-Ritz kernels: two PMC passes of scripts/pmc.py work + the report
    rm -rf /tmp/pmcm1_$TAG /tmp/pmcm2_$TAG
    (cd /tmp && timeout -k 10 ${LIMIT:-400} rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d /tmp/pmcm1_$TAG -o run -- python3 $R/scripts/pmc.py work "$@" > $O/${TAG}_p1.log 2>&1) || { tail -5 $O/${TAG}_p1.log; exit 1; }
    (cd /tmp && timeout -k 10 ${LIMIT:-400} rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcm2_$TAG -o run -- python3 $R/scripts/pmc.py work "$@" > $O/${TAG}_p2.log 2>&1) || { tail -5 $O/${TAG}_p2.log; exit 1; }
    python3 scripts/pmc.py mfma_report /tmp/pmcm1_$TAG /tmp/pmcm2_$TAG > $O/${TAG}_mfma.json; rc=$?
    head -c 3000 $O/${TAG}_mfma.json; exit $rc ;;
  pmc2)   # HBM traffic of the hot kernels: two PMC passes (FETCH_SIZE, WRITE_SIZE) of scripts/pmc.py + the report
    rm -rf /tmp/pmcf_$TAG /tmp/pmcw_$TAG
    (cd /tmp && timeout -k 10 ${LIMIT:-500} rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf_$TAG -o run -- python3 $R/scripts/pmc.py work "$@" $O/${TAG}_work.json > $O/${TAG}_fetch.log 2>&1) || { tail -5 $O/${TAG}_fetch.log; exit 1; }
    (cd /tmp && timeout -k 10 ${LIMIT:-500} rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw_$TAG -o run -- python3 $R/scripts/pmc.py work "$@" $O/${TAG}_work.json > $O/${TAG}_write.log 2>&1) || { tail -5 $O/${TAG}_write.log; exit 1; }
    python3 scripts/pmc.py report /tmp/pmcf_$TAG /tmp/pmcw_$TAG $O/${TAG}_work.json > $O/${TAG}_traffic.json; rc=$?
    python3 - $O/${TAG}_traffic.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j.items():
    if isinstance(v, dict) and "traffic_over_algorithmic" in v:
        print("%-28s read %.0f MB write %.0f MB  traffic/algorithmic %.3f" % (k, v["read_bytes_corrected"] / 1e6, v["write_bytes_corrected"] / 1e6, v["traffic_over_algorithmic"]))
    elif k == "calibration":
        print("calibration", v)
PY
    exit $rc ;;
  *) echo "unknown task $task"; exit 2 ;;
esac

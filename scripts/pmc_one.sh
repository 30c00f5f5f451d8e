#!/bin/bash
# one rocprofv3 --pmc pass (kernel trace only) of the SpMM micro-benchmark per counter name given; CSVs to gpurun_out/pmc1_<tag>_<counter>.csv
#   gpurun -- 'bash scripts/pmc_one.sh <tag> <counter> [<counter> ...]'      (MATRIX / REPS / GENEO_* pass through)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
tag=$1; shift
for c in "$@"; do
  rm -rf /tmp/pmc1_$c
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc1_$c -o run -- python3 $R/scripts/spmm_bench.py 126 > $O/pmc1_${tag}_$c.log 2>&1) || { tail -3 $O/pmc1_${tag}_$c.log; exit 1; }
  f=$(find /tmp/pmc1_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc1_${tag}_$c.csv
done
python3 - $O $tag "$@" <<'PY'
import csv, sys, collections
o, tag = sys.argv[1], sys.argv[2]
for c in sys.argv[3:]:
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/pmc1_%s_%s.csv" % (o, tag, c))) if "k_spmm_sell" in r["Kernel_Name"]]
    print("%-40s launches %4d  mean %.1f" % (c, len(v), sum(v) / max(1, len(v))))
PY

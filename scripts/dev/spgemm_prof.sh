#!/bin/bash
# per-kernel times of the device sparse products on a multigrid-like chain, both numeric-pass forms
# (rocprofv3 kernel stats of scripts/dev/spgemm_check.py <n>); run through gpurun
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
n=${1:-100}
for form in hash scan; do
  rm -rf /tmp/spg_$form
  if [ $form = scan ]; then export GENEO_SPGEMM_SCAN_FILL=1; else unset GENEO_SPGEMM_SCAN_FILL; fi
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/spg_$form -o run -- python3 $R/scripts/dev/spgemm_check.py $n > $O/spg_$form.log 2>&1) || { tail -5 $O/spg_$form.log; exit 1; }
  f=$(find /tmp/spg_$form -name "*kernel_stats.csv" | head -1)
  cp $f $O/spg_${form}_stats.csv
  t=$(find /tmp/spg_$form -name "*kernel_trace.csv" | head -1)
  echo "== $form"; grep -E "spgemm" $O/spg_${form}_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/'
  python3 - $t <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spgemm" in r["Kernel_Name"]]
for r in rows:
    print("   %-28s %9.1f us" % (r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done

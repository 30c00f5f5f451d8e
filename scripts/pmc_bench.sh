#!/bin/bash
# One rocprofv3 --pmc pass (kernel trace only) of a bench.py run per counter; per kernel instantiation whose name contains
# <filter>: launches and the mean counter value, as JSON with the kernel-source hash.
#   gpurun -- 'bash scripts/pmc_bench.sh <tag> <filter> "<bench.py args>" <counter> [<counter> ...]'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
tag=$1; filt=$2; bargs=$3; shift 3
for c in "$@"; do
  rm -rf /tmp/pmcb_$c
  (cd /tmp && timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcb_$c -o run -- python3 $R/bench.py $bargs > $O/pmcb_${tag}_$c.log 2>&1) || { tail -3 $O/pmcb_${tag}_$c.log; exit 1; }
  f=$(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/pmcb_${tag}_$c.csv
done
python3 - $R $tag "$filt" "$bargs" "$@" > $O/pmcb_${tag}.json <<'PY'
import csv, sys, json, hashlib, os
root, tag, filt, bargs = sys.argv[1:5]
out = {"kernel_source_sha16": hashlib.sha256(open(os.path.join(root, "geneo4petsc_amd", "csrc", "backend_hip.hip"), "rb").read()).hexdigest()[:16],
       "command": "rocprofv3 --pmc <one counter per pass> --kernel-trace -- python3 bench.py " + bargs,
       "note": "means per launch; _sum counters are summed over the chip's instances, FETCH_SIZE / WRITE_SIZE in units of 1024 B "
               "(FETCH_SIZE under-reports wide streaming reads on gfx950: scripts/pmc.py calibrates it on a known stream)", "kernels": {}}
for c in sys.argv[5:]:
    acc = {}
    for r in csv.DictReader(open("/tmp/pmcb_%s_%s.csv" % (tag, c))):
        k = r["Kernel_Name"]
        if filt not in k:
            continue
        short = k.split("(")[0].replace("bk::", "").replace("void ", "").strip()
        acc.setdefault(short, []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["kernels"].setdefault(k, {"launches": len(v)})[c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
PY
head -c 2500 $O/pmcb_${tag}.json

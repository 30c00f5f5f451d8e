#!/bin/bash
# Runs several scripts/gpu.sh tasks in one gpurun call: "task args" strings separated by ';;'.  A task that was killed by
# its time limit (124 / 137) ends the sequence -- no further GPU step after a hang; ordinary failures do not.
#   gpurun -- 'bash scripts/gpu_seq.sh "TAG=a tests tests/test_gpu_kernels.py -m gpu -q ;; TAG=b bench --steps 1"'
R=${GRAFT_REPO_ROOT:-$(pwd)}
IFS=$'\n' read -r -d '' -a steps < <(echo "$1" | sed 's/;;/\n/g' && printf '\0')
worst=0
for st in "${steps[@]}"; do
  st=$(echo "$st" | sed 's/^ *//;s/ *$//')
  [ -z "$st" ] && continue
  echo "=== $st"
  eval "env $(echo "$st" | grep -oE '^([A-Z][A-Z0-9_]*=[^ ]+ )*') bash $R/scripts/gpu.sh $(echo "$st" | sed -E 's/^([A-Z][A-Z0-9_]*=[^ ]+ )*//')"
  rc=$?
  echo "=== exit $rc"
  [ $rc -gt $worst ] && worst=$rc
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit: stopping the sequence"; break; fi
done
exit $worst

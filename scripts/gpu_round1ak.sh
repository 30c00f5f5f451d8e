#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/ak_counters.txt 2>&1
grep -i -E "mfma|lds_bank|LDS_BANK|SQ_BUSY_CYCLES|SQ_WAVES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU" $O/ak_counters.txt | head -40

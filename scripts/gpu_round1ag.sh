#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 GENEO_BENCH_COMM=staged MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2
for N in 2 4; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2957$N \
     bench.py --gpus $N --steps 1 --warmup 1 --n-per-gpu 64 --no-cpu-baseline > $O/ag_staged_n$N.log 2>&1
  rc=$?; echo "bench staged N=$N exit $rc"; tail -1 $O/ag_staged_n$N.log | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print(j['n_gpus'], 'setup %.3f' % j['setup_s'], j['setup_breakdown_s'], 'solve %.3f' % j['solve_s'], 'its', j['iterations'], 'dimE', j['dimE'], j['untimed_step_with_hip_graphs_s'])"
  [ $rc -eq 0 ] || exit $rc
done

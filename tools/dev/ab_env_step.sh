#!/bin/bash
# same-box A/B of the config-5 step (bf16, sync loss, train-mode BatchNorm) with / without one environment switch:
#   gpurun -- 'bash tools/dev/ab_env_step.sh S2L_NO_CONV_BSTATS [rounds=3]'
set -u
V=$1; N=${2:-3}
for i in $(seq $N); do for e in 0 1; do
  if [ $e = 1 ]; then export $V=1; else unset $V; fi
  ms=$(timeout 200 python tools/bench_train.py 64 bf16 --sync=8 --trainbn 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$V=$e ms_per_step $ms"
done; done

#!/bin/bash
# GPU: stage times of one evaluation at small sizes (the persistent-launch range), RBF iso D = 8 as BASELINE configs[1]
R=${GRAFT_REPO_ROOT:-$PWD}
for n in ${1:-3072 4096 4608 5120}; do
  timeout 200 python $R/bench.py --n $n --d 8 --kind rbf --iso --steps 200 --warmup 20 --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('N=$n', d['ms_per_step'], d['stage_ms'], 'lml %.12g' % d['lml'], d.get('small_n_schedule'), d.get('persist_aborts'))"
done

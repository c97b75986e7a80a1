#!/bin/bash
# A/B of the persistent dataflow Cholesky against the launch-per-step schedule on ONE box: whole evaluations through the C-ABI
export TMPDIR=/tmp
for n in 1024 2048 3072 4096 5120; do
  for p in 0 1; do
    MI355GP_PERSIST=$p timeout 120 python bench.py --n $n --d 8 --kind rbf --iso --steps 300 --warmup 20 --no-legs --no-cpu-baseline --no-parity-gate --abi-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n persist=$p ms_per_step %.4f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
  done
done

#!/bin/bash
# Run on the GPU box: A/B of the paired far updates (MI355GP_AGG2) on one box.
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
for N in 16384 8192 32768; do
  D=32; KIND="--kind matern52"; ST=20
  if [ $N != 16384 ]; then D=8; KIND="--kind rbf --iso"; fi
  if [ $N == 32768 ]; then ST=4; fi
  for A in 0 1 0 1; do
    MI355GP_AGG2=$A timeout 300 python bench.py --n $N --d $D $KIND --steps $ST --warmup 2 --no-legs --no-cpu-baseline --no-parity-gate --abi-only 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip())
print('N=$N agg2=$A: %.3f ms' % d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()}, 'upd', d['roofline']['launches_per_step'], '%.1f us' % (1e3 * d['roofline']['avg_launch_ms']), 'frac %.3f' % d['roofline']['frac'])
"
  done
done

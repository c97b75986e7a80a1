#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_grid.py tests/test_gpu_baseline.py -m gpu -q --maxfail=8 --durations=5 ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
for la in 1 0; do
  for cfg in "1x1 16384" "1x1 32768" "2x2 16384" "2x4 32768"; do
    set -- $cfg
    MI355GP_GRID_LOOKAHEAD=$la timeout 300 python bench.py --grid $1 --n $2 --d 8 --kind rbf --iso --steps 3 --warmup 1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('la=$la', '$1', d['config']['N'], 'ms', round(d['ms_per_step'],1), 'frac', round(d['iteration_frac_of_fp64_peak'],3), d['stage_ms'], d.get('parity_vs_golden'))
    elif 'rror' in l: print(l.strip()[:300])
"
  done
done 2>&1 | tee $O/grid.log
python tools/sweep_env.py MI355GP_TRI64_MAX unset --n 2048,4096,8192,16384,32768 --reps 3 --full 2>&1 | tee $O/single.log

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
( python tools/sweep_env.py MI355GP_PART1_ON_PANEL 0,1,0,1 --n 4096,6144,16384 --reps 3 --full ) > $O/sweep9.log 2>&1
cat $O/sweep9.log | cut -c1-150
for v in 0 1; do
MI355GP_PART1_ON_PANEL=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-grid-leg --no-cpu-baseline 2>>$O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p1=$v C3 drop-in', round(d['ms_per_step'],3), d['stage_ms'])"
done

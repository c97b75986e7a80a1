#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
( python tools/sweep_env.py MI355GP_PART1_ON_PANEL 0,1,2,0,1,2 --n 8192,16384 --reps 3 --full ) > $O/sweep10.log 2>&1
cat $O/sweep10.log | cut -c1-150
for v in 1 2; do
MI355GP_PART1_ON_PANEL=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-grid-leg --no-cpu-baseline --no-parity-gate 2>>$O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p1=$v C3', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['cholesky_gflops'])"
done

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
for t in 48 80 48 80; do
for n in 6144 8192 10112; do
MI355GP_TRI_MIN_NT=$t timeout 300 python bench.py --n $n --d 8 --kind rbf --iso --steps 60 --warmup 10 --no-grid-leg --no-cpu-baseline --no-parity-gate 2>>$O/bench2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('min_nt=$t n=$n', round(d['ms_per_step'],3), d['stage_ms']['total'])"
done; done
tail -3 $O/bench2.err

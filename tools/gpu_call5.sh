#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_baseline.py -m gpu -q --maxfail=10 --durations=5 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log
timeout 300 python bench.py --sparse --steps 5 --warmup 2 2>&1 | tail -2 | tee $O/sparse_bench.json
for nb in 256 512 1024; do
MI355GP_GRID_LOOKAHEAD=1 timeout 300 python bench.py --grid 1x1 --n 16384 --d 8 --kind rbf --iso --steps 3 --warmup 1 --nb $nb 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('nb=$nb 1x1', d['config']['N'], 'ms', round(d['ms_per_step'],1), d['stage_ms'])
"
done | tee $O/grid_nb.log

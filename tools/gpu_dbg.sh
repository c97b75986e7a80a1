#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=5 ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
for g in 0 1 0 1; do
for n in 2048 4096; do
MI355GP_GRAPH=$g timeout 300 python bench.py --n $n --d 8 --kind rbf --iso --steps 200 --warmup 10 --no-grid-leg --no-cpu-baseline --no-parity-gate 2>>$O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph=$g n=$n', round(d['ms_per_step'],3), d['host_path'], d['stage_ms']['total'])"
done; done
MI355GP_GRAPH=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-grid-leg --no-cpu-baseline 2>>$O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', round(d['ms_per_step'],3), d['stage_ms'], d['parity'])"
tail -3 $O/bench.err

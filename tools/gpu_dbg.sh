#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x ) > $O/pytest_sparse.log 2>&1
tail -3 $O/pytest_sparse.log
for v in 0 1 0 1; do
MI355GP_SPARSE_KMM_OVERLAP=$v timeout 300 python bench.py --sparse --steps 10 --warmup 3 --no-cpu-baseline 2>>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap=$v', round(d['ms_per_step'],3), d['stage_ms'])"
done
tail -3 $O/err.log

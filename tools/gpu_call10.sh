#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_grid.py tests/test_gpu_baseline.py -m gpu -q --maxfail=10 -k "grid or config4" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/sweep_env.py MI355GP_NBO 512,768,1024 --n 16384,32768 --reps 2 --full 2>&1 | tee $O/nbo_big.log
timeout 300 python bench.py --grid 1x1 --n 32768 --d 8 --kind rbf --iso --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-600

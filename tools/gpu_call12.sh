#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline.py -m gpu -q -x 2>&1 | tail -5
  python tools/sweep_env.py MI355GP_PANEL_NEXT 0,1,0,1 --n 2048,4096,8192 --full
  python tools/sweep_env.py MI355GP_PANEL_NEXT 0,1 --n 16384 --reps 3 --full
) > $O/sweep4.log 2>&1
cat $O/sweep4.log | cut -c1-200

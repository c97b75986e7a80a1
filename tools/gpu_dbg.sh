#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
( python tools/sweep_env.py MI355GP_LDS_SLOT 0,1,0,1 --n 4096,8192,16384 --reps 3 --full
  python tools/sweep_env.py MI355GP_LDS_SLOT 0,1 --n 32768 --reps 2
) > $O/sweep6.log 2>&1
cat $O/sweep6.log | cut -c1-180

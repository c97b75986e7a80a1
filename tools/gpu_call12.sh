#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
( python tools/sweep_env.py MI355GP_XCD_TILES 0,1,0,1 --n 8192,16384 --reps 3 --full
  python tools/sweep_env.py MI355GP_XCD_TILES 0,1 --n 32768 --reps 2
) > $O/sweep5.log 2>&1
cat $O/sweep5.log | cut -c1-200

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
( python tools/sweep_env.py MI355GP_SPLIT_PART1 0,1,0,1 --n 2048,4096,8192 --full
  python tools/sweep_env.py MI355GP_SPLIT_PART1 0,1 --n 16384 --reps 3 --full
) > $O/sweep3.log 2>&1
cat $O/sweep3.log | cut -c1-200

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
( python tools/sweep_env.py MI355GP_PART1_ON_PANEL 1,2,1,2 --n 6144,8192,16384 --reps 3 --full
  python tools/sweep_env.py MI355GP_PART1_ON_PANEL 1,2 --n 32768 --reps 2 ) > $O/sweep11.log 2>&1
cat $O/sweep11.log | cut -c1-150

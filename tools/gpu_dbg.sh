#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
( python tools/sweep_env.py MI355GP_NBO 512,384,640,1024 --n 16384 --reps 3 --full
  python tools/sweep_env.py MI355GP_NBO 512,384,1024 --n 8192 --reps 3 --full
  python tools/sweep_env.py MI355GP_TRI_MIN_NT 48,32 --n 4096,5120 --reps 3 --full ) > $O/sweep12.log 2>&1
cat $O/sweep12.log | cut -c1-150

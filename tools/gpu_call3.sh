#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=8 --durations=8 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
python tools/sweep_env.py MI355GP_TRI64_MAX 0,64,256,512,1024 --n 2048,4096,6144,8192 --full 2>&1 | tee $O/tri64.log
python tools/sweep_env.py MI355GP_LAUUM64_MAX 0,136,528,1176,2080 --n 2048,4096,6144,8192 --full 2>&1 | tee $O/lauum64.log

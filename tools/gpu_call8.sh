#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse.py tests/test_gpu_sum.py -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/sweep_env.py MI355GP_TRI64_MAX unset --n 2048,4096,8192,16384 --reps 3 --full 2>&1 | tee $O/single.log

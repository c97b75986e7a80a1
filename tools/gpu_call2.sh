#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline.py tests/test_gpu_sum.py -m gpu -q --maxfail=8 ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python tools/sweep_env.py MI355GP_UPD64_MAX 0,64,128,192,256,384,512 --n 2048,4096,8192,16384 --full 2>&1 | tee $O/upd64.log

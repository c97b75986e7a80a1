#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_predict.py tests/test_gpu_parity.py tests/test_gpu_sparse.py -m gpu -q -x ) > $O/pytest_predict.log 2>&1
tail -30 $O/pytest_predict.log

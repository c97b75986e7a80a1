#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log

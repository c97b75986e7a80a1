#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x -k "integration_stub" 2>&1 | tail -25 ) > $O/pytest_sp.log 2>&1
cat $O/pytest_sp.log

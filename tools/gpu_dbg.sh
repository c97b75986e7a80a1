#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_reference_tests.py -m gpu -q -x 2>&1 | tail -40 ) > $O/pytest_reft.log 2>&1
cat $O/pytest_reft.log

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
timeout 600 python examples/gp_regression.py 4096 4 > $O/example.log 2>&1; echo rc=$?; tail -12 $O/example.log

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "graph_replay" 2>&1 | tail -15 ) > $O/pytest_graph.log 2>&1
cat $O/pytest_graph.log

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
timeout 600 python tools/time_callers.py 2>$O/tc.err | tee $O/time_callers.json; tail -3 $O/tc.err

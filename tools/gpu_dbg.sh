#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
for n in 4096 16384; do timeout 900 python tools/vendor_compare.py --n $n 2>$O/err_$n.log | tee $O/vendor_$n.json; tail -2 $O/err_$n.log; done

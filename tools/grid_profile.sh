#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats of the block-cyclic grid mode over the loopback transport.
#   tools/grid_profile.sh TAG [N] [PrxPc]
TAG=${1:-grid}
N=${2:-32768}
GRID=${3:-2x4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --grid $GRID --n $N --d 8 --kind rbf --iso --steps 2 --warmup 1 --grid-child --device 0"
timeout 400 $CMD > $OUT/plain.log 2>&1
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1 || echo "rc $?"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -3 $OUT/plain.log | cut -c1-1500
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-170

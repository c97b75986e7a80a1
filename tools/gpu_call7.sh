#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $O
cd /tmp
for n in 4096 8192; do
rocprofv3 --output-format csv --kernel-trace -d $O/tr$n -o run -- python $GRAFT_REPO_ROOT/bench.py --n $n --d 8 --kind rbf --iso --steps 3 --warmup 2 --no-cpu-baseline --no-grid-leg --no-parity-gate --abi-only > $O/tr$n.log 2>&1
python $GRAFT_REPO_ROOT/tools/chain_trace.py $O/tr$n/run_kernel_trace.csv --show 60 > $O/chain$n.txt 2>&1
head -30 $O/chain$n.txt
done
rocprofv3 --output-format csv --kernel-trace -d $O/tr16384 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-grid-leg --no-parity-gate --abi-only > $O/tr16384.log 2>&1
python $GRAFT_REPO_ROOT/tools/potrf_timeline.py $O/tr16384/run_kernel_trace.csv > $O/timeline16384.txt 2>&1
python $GRAFT_REPO_ROOT/tools/chain_trace.py $O/tr16384/run_kernel_trace.csv > $O/chain16384.txt 2>&1
cat $O/chain16384.txt | head -24; cat $O/timeline16384.txt | head -40
find $O -name "*.csv" -size +8M -delete; find $O -name "*.db" -delete

#!/bin/bash
# GPU: kernel timeline of ONE evaluation at a small size (default N = 4096 = BASELINE configs[1]) from a rocprofv3 kernel trace
#   bash tools/c2_trace.sh OUTDIR_UNDER_gpurun_out [N]
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; N=${2:-4096}; mkdir -p $O
rm -rf /tmp/c2tr; timeout 300 rocprofv3 --output-format csv --kernel-trace -d /tmp/c2tr -o run -- python $R/bench.py --n $N --d 8 --kind rbf --iso --steps 40 --warmup 10 --no-legs --no-cpu-baseline --no-parity-gate > $O/trace_bench_$N.json 2> $O/trace_prof_$N.err
f=$(find /tmp/c2tr -name "run_kernel_trace.csv" | head -1); for b in 8 20; do python $R/tools/eval_timeline.py $f $b; done > $O/timeline_$N.txt

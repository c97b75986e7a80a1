#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + the three PMC passes of bench.py's default workload, and the
# sparse workload.  Outputs under gpurun_out/prof_$TAG; summarise with tools/summarize_profile.py and commit profiles/.
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-grid-leg --no-parity-gate --abi-only"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
CMD1="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grid-leg --no-parity-gate --abi-only"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_a -o run -- $CMD1 > $OUT/pmc_a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_b -o run -- $CMD1 > $OUT/pmc_b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c -o run -- $CMD1 > $OUT/pmc_c.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/sparse_stats -o run -- python $PWD/bench.py --sparse --steps 3 --warmup 1 > $OUT/sparse_stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/sparse_pmc_b -o run -- python $PWD/bench.py --sparse --steps 1 --warmup 1 > $OUT/sparse_pmc_b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/sparse_pmc_c -o run -- python $PWD/bench.py --sparse --steps 1 --warmup 1 > $OUT/sparse_pmc_c.log 2>&1
find $OUT -name "*.csv" | head -40
du -sh $OUT

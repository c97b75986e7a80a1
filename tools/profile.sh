#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + the PMC passes (separate runs, kernel-trace only) of the SHIPPED
# schedule for BASELINE configs[2] (C3, the default bench.py workload), configs[1] (C2) and configs[4] (C5, --sparse).
# Outputs under gpurun_out/prof_$TAG; summarise with tools/summarize_profile.py and commit profiles/$TAG_*.
TAG=${1:-r4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
run() {   # run <subdir> <rocprof args...> -- <cmd...>
    local sub=$1; shift
    timeout -k 5 150 rocprofv3 --output-format csv --kernel-trace "$@" > $OUT/$sub.log 2>&1 || echo "$sub: rc $?"
}
C3="python $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-parity-gate --abi-only"
C3ONE="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs --no-parity-gate --abi-only"
C2="python $PWD/bench.py --n 4096 --d 8 --kind rbf --iso --steps 20 --warmup 3 --no-cpu-baseline --no-legs --no-parity-gate --abi-only"
C2ONE="python $PWD/bench.py --n 4096 --d 8 --kind rbf --iso --steps 2 --warmup 3 --no-cpu-baseline --no-legs --no-parity-gate --abi-only"
C5="python $PWD/bench.py --sparse --steps 3 --warmup 1 --no-cpu-baseline"
C5ONE="python $PWD/bench.py --sparse --steps 1 --warmup 1 --no-cpu-baseline"
# PROFILE_SET=mini: the three kernel-stats runs and the SQ counter pass of the headline only
# PROFILE_SET=lite: the three kernel-stats runs, the PMC passes of the headline (C3) and of the sparse path (C5) only
LITE=${PROFILE_SET:-full}
run stats --stats -d $OUT/stats -o run -- $C3
run pmc_a --pmc $PMC_SQ -d $OUT/pmc_a -o run -- $C3ONE
if [ "$LITE" != mini ]; then
run pmc_b --pmc FETCH_SIZE -d $OUT/pmc_b -o run -- $C3ONE
run pmc_c --pmc WRITE_SIZE -d $OUT/pmc_c -o run -- $C3ONE
fi
run c2_stats --stats -d $OUT/c2_stats -o run -- $C2
if [ "$LITE" != lite ] && [ "$LITE" != mini ]; then
run c2_pmc_a --pmc $PMC_SQ -d $OUT/c2_pmc_a -o run -- $C2ONE
run c2_pmc_b --pmc FETCH_SIZE -d $OUT/c2_pmc_b -o run -- $C2ONE
run c2_pmc_c --pmc WRITE_SIZE -d $OUT/c2_pmc_c -o run -- $C2ONE
fi
run sparse_stats --stats -d $OUT/sparse_stats -o run -- $C5
if [ "$LITE" != lite ] && [ "$LITE" != mini ]; then
run sparse_pmc_a --pmc $PMC_SQ -d $OUT/sparse_pmc_a -o run -- $C5ONE
fi
if [ "$LITE" != mini ]; then
run sparse_pmc_b --pmc FETCH_SIZE -d $OUT/sparse_pmc_b -o run -- $C5ONE
run sparse_pmc_c --pmc WRITE_SIZE -d $OUT/sparse_pmc_c -o run -- $C5ONE
fi
# the kernel traces are large: keep the stats tables and the counter tables only
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*.csv" | head -60
du -sh $OUT

#!/bin/bash
# First run on a node with MORE THAN ONE GPU (none was available to the builder or the driver through round 5): every step that
# only RCCL itself can exercise, in the order that isolates a failure, each with its own timeout and a one-line PASS / FAIL.
#   tools/rccl_first_light.sh [ngpus=all visible]
# 1  transport self-test over RCCL, 2 ranks (1x2): ncclCommInitRank, ncclCommSplit colour / key numbering, several ncclBroadcasts
#    with different roots inside ONE ncclGroupStart / End, out-of-place broadcast on the root, all-reduces (mi355gp_dbg_comm_selftest)
# 2  the same on the full grid shape of the node (2x2 / 2x4)
# 3  the two GPU tests that need two devices (skipped on the 1-GPU boxes): tests/test_gpu_sparse.py::test_two_process_rccl_over_xgmi[grid|sparse]
# 4  1x2 and 2x2 block-cyclic grids at N=4096 over RCCL with the collective-sequence check on, against the loopback bits
# 5  the 2-way row-sharded VarDTC over RCCL against the unsharded run
# 6  bench.py --gpus N (replicas + the grid / c5 legs over all ranks) and bench.py --grid PrxPc --gpus N
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=${TMPDIR:-/tmp}
unset MI355GP_TRANSPORT MI355GP_BENCH_BACKEND
NG=${1:-$(python -c "from gpy_amd import _lib as L; print(L.device_count())")}
O=gpurun_out/first_light
mkdir -p $O
if [ "$NG" -lt 2 ]; then echo "FAIL  needs at least two GPUs (found $NG)"; exit 2; fi
case "$NG" in 2|3) SHAPE=1x2; W=2 ;; 4|5|6|7) SHAPE=2x2; W=4 ;; *) SHAPE=2x4; W=8 ;; esac
PORT=$((29700 + RANDOM % 200))
step() {   # step <name> <timeout_s> <command...>
    local name=$1 t=$2; shift 2
    if timeout -k 10 "$t" "$@" > "$O/$name.log" 2>&1; then echo "PASS  $name"; else echo "FAIL  $name (rc $?, see $O/$name.log)"; tail -5 "$O/$name.log" | cut -c1-300; fi
}
selftest() {   # selftest <Pr> <Pc>: one process per rank, rank r on device r
    local Pr=$1 Pc=$2 w=$(( $1 * $2 )) d=$(mktemp -d) rc=0 pids=()
    for r in $(seq 0 $((w - 1))); do
        RANK=$r WORLD_SIZE=$w LOCAL_RANK=$r MI355GP_ID_DIR=$d MI355GP_JOB_NONCE=fl$$ python -c "
from gpy_amd import grid as G
bad, cs, rrow, rcol = G.comm_selftest($Pr, $Pc, count=1 << 20, rounds=9)
print('rank', $r, 'mismatches', bad, 'row rank', rrow, 'col rank', rcol)
assert bad == 0 and (rrow, rcol) == ($r % $Pc, $r // $Pc)" &
        pids+=($!)
    done
    for p in "${pids[@]}"; do wait $p || rc=1; done
    rm -rf "$d"
    return $rc
}
export -f selftest
step 1_selftest_1x2 120 bash -c "selftest 1 2"
step 2_selftest_$SHAPE 180 bash -c "selftest ${SHAPE%x*} ${SHAPE#*x}"
step 3_two_gpu_tests 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_grid.py -m gpu -q -k "two_process_rccl_over_xgmi or rccl_transport" -x
for shape in 1x2 2x2; do
    w=$(( ${shape%x*} * ${shape#*x} ))
    [ "$w" -le "$NG" ] || continue
    step 4_grid_${shape}_n4096 300 env MI355GP_GRID_CHECK_SEQ=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 \
        --master-port $((PORT + w)) bench.py --grid $shape --gpus $w --size 4096 --dims 8 --kind rbf --iso --steps 2 --warmup 1
done
step 5_sparse_2way 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((PORT + 20)) \
    bench.py --sparse --gpus 2 --size 40000 --inducing 512 --steps 2 --warmup 1 --no-cpu-baseline
step 6_bench_gpus_$W 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((PORT + 40)) \
    bench.py --gpus $W --steps 10 --warmup 3
step 6_bench_grid_$SHAPE 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((PORT + 60)) \
    bench.py --grid $SHAPE --gpus $W --size 32768 --dims 8 --kind rbf --iso --steps 3 --warmup 1
grep -h "^{" $O/6_bench_*.log | cut -c1-600

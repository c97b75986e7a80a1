#!/bin/bash
# Dry run of the MULTI-RANK flows of bench.py on a box with ONE GPU: N processes under torch.distributed.run, rank plumbing over
# gloo (MI355GP_BENCH_BACKEND), every rank on device 0, and the data-path collectives of the block-cyclic grid leg and of the
# row-sharded sparse leg over the hipIpc transport (MI355GP_TRANSPORT=ipc, csrc/ipc_comm.hip) that stands in for RCCL, which
# refuses two ranks on one device.  What it exercises before an 8-GPU node ever sees the code: RANK / LOCAL_RANK / WORLD_SIZE
# handling, the id exchange, ncclCommSplit-style row / column communicators, the per-rank code of csrc/grid.hip with
# g->ranks.size() == 1, the child legs with their timeouts, the one JSON line assembled by rank 0.
#   tools/multiproc_dryrun.sh [ranks=8] [full]      "full": BASELINE sizes in every leg (minutes); default: small legs
set -u
cd "$(dirname "$0")/.."
N=${1:-8}
MODE=${2:-small}
export MI355GP_BENCH_BACKEND=gloo MI355GP_TRANSPORT=ipc HSA_ENABLE_IPC_MODE_LEGACY=0 MI355GP_GRID_CHECK_SEQ=1
export MI355GP_IPC_TIMEOUT_S=${MI355GP_IPC_TIMEOUT_S:-120}
PORT=$((29600 + RANDOM % 300))
mkdir -p gpurun_out
if [ "$MODE" = full ]; then
    ARGS="--gpus $N --steps 3 --warmup 1"
else
    ARGS="--gpus $N --steps 2 --warmup 1 --size 4096 --dims 8 --kind rbf --iso --grid-n 8192 --dry-run-sizes"
fi
echo "== default bench line with $N ranks on one GPU ($MODE legs)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
    bench.py $ARGS | tee gpurun_out/dryrun_bench_${N}.json | tail -c 3000
echo
echo "== --grid mode with $N ranks: one problem on the block-cyclic grid"
case "$N" in 2) SHAPE=1x2 ;; 4) SHAPE=2x2 ;; 8) SHAPE=2x4 ;; *) SHAPE=1x$N ;; esac
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 50)) \
    bench.py --grid "$SHAPE" --gpus "$N" --size 8192 --dims 8 --kind rbf --iso --steps 2 --warmup 1 | tee gpurun_out/dryrun_grid_${N}.json | tail -c 1500

#!/bin/bash
# The two multi-rank modes of bench.py alone (no child legs), N ranks sharing ONE GPU over the hipIpc transport: a quick check of
# the rank plumbing (seconds instead of the minutes of tools/multiproc_dryrun.sh).   tools/multiproc_quick.sh [ranks=4]
cd "$(dirname "$0")/.."
N=${1:-4}
export MI355GP_BENCH_BACKEND=gloo MI355GP_TRANSPORT=ipc HSA_ENABLE_IPC_MODE_LEGACY=0 MI355GP_GRID_CHECK_SEQ=1 MI355GP_IPC_TIMEOUT_S=60
PORT=$((29300 + RANDOM % 200))
case "$N" in 2) SHAPE=1x2 ;; 4) SHAPE=2x2 ;; 8) SHAPE=2x4 ;; *) SHAPE=1x$N ;; esac
echo "== --sparse, rows sharded over $N ranks"
timeout -k 5 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
    bench.py --sparse --gpus "$N" --size 20000 --inducing 512 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -c 2500
echo
echo "== --grid $SHAPE"
timeout -k 5 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 31)) \
    bench.py --grid "$SHAPE" --gpus "$N" --size 4096 --dims 8 --kind rbf --iso --steps 2 --warmup 1 2>&1 | tail -c 2500

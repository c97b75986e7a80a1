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
    bench.py $ARGS > gpurun_out/dryrun_bench_${N}.json 2> gpurun_out/dryrun_bench_${N}.err
echo "rc $?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids\|OMP_NUM\|\*\*\*\*" gpurun_out/dryrun_bench_${N}.err | head -40 | cut -c1-400
grep "^{" gpurun_out/dryrun_bench_${N}.json | tail -1 | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
    print({k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step', 'scaling')})
    for leg in ('c2', 'grid', 'grid16k', 'grid4k', 'c5', 'c4_single'):
        r = d.get(leg) or {}
        print(' ', leg, {k: r.get(k) for k in ('ms_per_step', 'n_gpus', 'N', 'transport', 'stage_ms_rank_min_max', 'error', 'stderr_tail', 'parity_vs_golden') if r.get(k) is not None})
except Exception as e:
    print('no JSON line:', e)
"
echo
echo "== --grid mode with $N ranks: one problem on the block-cyclic grid"
case "$N" in 2) SHAPE=1x2 ;; 4) SHAPE=2x2 ;; 8) SHAPE=2x4 ;; *) SHAPE=1x$N ;; esac
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 50)) \
    bench.py --grid "$SHAPE" --gpus "$N" --size 8192 --dims 8 --kind rbf --iso --steps 2 --warmup 1 > gpurun_out/dryrun_grid_${N}.json 2> gpurun_out/dryrun_grid_${N}.err
echo "rc $?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids\|OMP_NUM\|\*\*\*\*" gpurun_out/dryrun_grid_${N}.err | head -20 | cut -c1-400
grep "^{" gpurun_out/dryrun_grid_${N}.json | tail -1 | cut -c1-900

python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --sparse --steps 2 --warmup 1 2>&1 | tail -2 | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 1 --grid 1x1 --n 4096 --d 8 --kind rbf --iso --steps 2 --warmup 1 2>&1 | tail -2 | cut -c1-300

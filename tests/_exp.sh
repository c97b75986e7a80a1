MI355GP_UPDATE_V2=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid.py -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error" | head -3
for v in 0 1; do echo "== UPDATE_V2=$v"; MI355GP_UPDATE_V2=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'], d['roofline']['achieved'], d['lml'])"; done
for v in 0 1; do echo "== N=4096 UPDATE_V2=$v"; MI355GP_UPDATE_V2=$v python bench.py --n 4096 --d 8 --kind rbf --iso --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms']['potrf'], d['stage_ms']['total'], d['lml'])"; done

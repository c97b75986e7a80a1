MI355GP_REVERSE_K=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -a "passed\|failed"
for cfg in "0" "1"; do echo "== REV=$cfg"; MI355GP_REVERSE_K=$cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])"; done

python tests/gemm_probe.py 4096x4096x4096 | cut -c1-130
python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'], d['roofline']['achieved'], d['lml'])"

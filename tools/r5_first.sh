#!/bin/bash
# round 5 GPU call: the new 128 x 128 factor -- correctness first, then the chain timeline and the two small benches
export TMPDIR=/tmp
O=gpurun_out/${1:-r5b}
mkdir -p $O
tools/hwprobe/diag128_probe > $O/diag128_probe.log 2>&1; cat $O/diag128_probe.log
timeout 300 python tools/diag_check.py > $O/diag_check.log 2>&1; tail -2 $O/diag_check.log
timeout 300 python tools/persist_probe.py 1024,2048,4096 > $O/persist_probe.log 2>&1; grep -E "^N=|per chain" $O/persist_probe.log
for p in 0 1; do
MI355GP_PERSIST=$p timeout 120 python bench.py --n 4096 --d 8 --kind rbf --iso --steps 300 --warmup 20 --no-legs --no-cpu-baseline --no-parity-gate --abi-only 2>$O/c2_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=4096 persist=$p ms_per_step %.4f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
done
timeout 200 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline --abi-only 2>$O/c3.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 ms_per_step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()}, d.get('families'))"
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -x ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3

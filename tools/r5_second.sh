#!/bin/bash
# round 5 GPU call: early inverse underneath the persistent launch (A/B), the part-2 queue bounding experiment, C5, full suite
export TMPDIR=/tmp
O=gpurun_out/${1:-r5d}
mkdir -p $O
for n in 2048 4096; do
for p in 0 1 0 1; do
MI355GP_PERSIST_TRI=$p timeout 120 python bench.py --n $n --d 8 --kind rbf --iso --steps 300 --warmup 20 --no-legs --no-cpu-baseline --no-parity-gate --abi-only 2>$O/c2_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n persist_tri=$p ms_per_step %.4f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
done
done
timeout 300 python tools/upd_queue_probe.py 16384 32768 > $O/upd_queue_probe.log 2>&1; cat $O/upd_queue_probe.log
timeout 200 python bench.py --sparse --steps 12 --warmup 3 --no-cpu-baseline 2>$O/c5.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 ms_per_step %.3f' % d['ms_per_step'], d['stage_ms'], d['roofline']['frac'], d['roofline']['k_gram_splitk'])"
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -x ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3

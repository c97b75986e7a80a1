#!/bin/bash
# same-box A/B through the DRIVER's command (drop-in classes, parity gate on): round-4 tree vs the working tree
export TMPDIR=/tmp
for rep in 1 2 3; do
for t in r4 HEAD; do
    d=.; [ $t = r4 ] && d=.ab/r4
    ( cd $d && timeout 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t run $rep: ms_per_step %.3f  stage total %.3f  potrf %.2f trtri %.2f lauum %.2f  abi %.3f' % (d['ms_per_step'], d['stage_ms']['total'], d['stage_ms']['potrf'], d['stage_ms']['trtri'], d['stage_ms']['lauum'], d['host_path']['abi_ms_per_step']))"
done
done

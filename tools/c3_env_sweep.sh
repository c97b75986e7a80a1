#!/bin/bash
# Same-box sweep of the process-default schedule switches at the headline configuration (C3: Matern-5/2 ARD N=16384 D=32, bare
# C-ABI, 20 steps): one line per setting, baseline first and last.   tools/c3_env_sweep.sh "VAR=val" "VAR=val VAR2=val" ...
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
export TMPDIR=/tmp
run() {
    ( env $1 timeout 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --abi-only --no-parity-gate 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['stage_ms']
print('%-44s ms_per_step %.3f  total %.3f  potrf %.2f trtri %.2f lauum %.2f' % ('${1:-default}', d['ms_per_step'], s['total'], s['potrf'], s['trtri'], s['lauum']))"
}
run "MI355GP_NOP=0"
for s in "$@"; do run "$s"; done
run "MI355GP_NOP=0"

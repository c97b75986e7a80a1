#!/bin/bash
# Run on the GPU box: the loopback block-cyclic grid (bench.py --grid) over group sizes.  tools/grid_sweep.sh N PrxPc "G:GW ..."
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
N=${1:-32768}
GRID=${2:-2x4}
SETS=${3:-"1:1 2:0 4:0 8:0 4:8"}
for s in $SETS; do
  G=${s%%:*}; GW=${s##*:}
  MI355GP_GRID_G=$G MI355GP_GRID_GW=$GW MI355GP_GRID_FORCE_GENERIC=1 timeout 300 python bench.py --grid $GRID --n $N --d 8 --kind rbf --iso \
     --steps 2 --warmup 1 --grid-child --device 0 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print('G=$G GW=$GW N=$N $GRID: %.1f ms' % d['ms_per_step'], d.get('stage_ms'), d.get('parity_vs_golden', {}).get('lml_rel'))
except Exception as e:
    print('G=$G GW=$GW failed:', l[-300:])
"
done

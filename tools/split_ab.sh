#!/bin/bash
# A/B of the split hand-over of the sub-diagonal tiles in the persistent Cholesky (persist.hip: the owner of tile (i, i-2)
# applies the last column to tile (i, i-1) itself; MI355GP_PERSIST_TUNE=4 switches it off): bit-identity tests, the chain's
# per-step timeline, whole evaluations.  ONE box.
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/split_ab}
mkdir -p "$OUT"
timeout -k 5 200 python -m pytest tests/test_gpu_linalg.py -q -x -k "persist" 2>&1 | tail -4
for T in 0 4 0 4; do
  echo "== tune $T"
  MI355GP_PERSIST_TUNE=$T timeout -k 5 60 python tools/persist_probe.py 1024,2048,4096 2>&1 | grep -v "^   step"
done
MI355GP_PERSIST_TUNE=0 timeout -k 5 60 python tools/persist_probe.py 4096 2>&1 | grep "^   step"
for T in 0 4 0 4; do
  MI355GP_PERSIST_TUNE=$T timeout -k 5 120 python bench.py --n 4096 --d 8 --kind rbf --iso --steps 300 --warmup 20 --no-legs --no-cpu-baseline \
    > "$OUT/c2_t$T.json" 2> "$OUT/c2_t$T.err"
  python - "$OUT/c2_t$T.json" $T <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("c2 tune=%s  %.4f ms/step  stage %s" % (sys.argv[2], d["ms_per_step"], d["stage_ms"]))
except Exception as e:
    print("c2 tune=%s FAILED %r" % (sys.argv[2], e))
PY
done

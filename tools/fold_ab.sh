#!/bin/bash
# A/B of the folded persistent launch (MI355GP_PERSIST=2: potrf + inverse + X^T X in one dataflow launch) against the
# factorisation-only persistent launch (1) and the launch-per-step schedule (0), whole evaluations on ONE box.
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/fold_ab}
mkdir -p "$OUT"
for N in 1024 2048 4096; do
  for P in 2 1 2 1; do
    MI355GP_PERSIST=$P timeout 300 python bench.py --n $N --d 8 --kind rbf --iso --steps 200 --warmup 20 --no-legs --no-cpu-baseline \
      > "$OUT/n${N}_p${P}.json" 2> "$OUT/n${N}_p${P}.err"
    python - "$OUT/n${N}_p${P}.json" $N $P <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("N=%s persist=%s  %.4f ms/step  stage %s  aborts? lml %.10f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["stage_ms"], d["lml"]))
except Exception as e:
    print("N=%s persist=%s FAILED %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
for P in 2 1; do
  MI355GP_PERSIST=$P timeout 300 python bench.py --sparse --steps 12 --warmup 3 --no-cpu-baseline > "$OUT/sparse_p${P}.json" 2> "$OUT/sparse_p${P}.err"
  python - "$OUT/sparse_p${P}.json" $P <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("sparse persist=%s  %.3f ms/step  stage %s" % (sys.argv[2], d["ms_per_step"], d["stage_ms"]))
except Exception as e:
    print("sparse persist=%s FAILED %r" % (sys.argv[2], e))
PY
done

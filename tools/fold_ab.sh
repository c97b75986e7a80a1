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
for CFG in "2 1" "2 0" "1 1" "2 1"; do
  set -- $CFG
  P=$1; K=$2
  MI355GP_PERSIST=$P MI355GP_KBUILD_STRIP=$K timeout 300 python bench.py --sparse --steps 12 --warmup 3 --no-cpu-baseline > "$OUT/sparse_p${P}_k${K}.json" 2> "$OUT/sparse_p${P}_k${K}.err"
  python - "$OUT/sparse_p${P}_k${K}.json" $P $K <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("sparse persist=%s kbuild_strip=%s  %.3f ms/step  stage %s  parity %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["stage_ms"], d.get("parity_checked")))
except Exception as e:
    print("sparse persist=%s strip=%s FAILED %r" % (sys.argv[2], sys.argv[3], e))
PY
done

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3
for n in 2048 4096; do timeout 300 python bench.py --n $n --d 8 --kind rbf --iso --steps 100 --warmup 10 --no-grid-leg --no-cpu-baseline > $O/bench_n$n.json 2>> $O/bench.err; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-grid-leg --no-cpu-baseline > $O/bench.json 2>> $O/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2k/bench*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d.get("stage_ms"), d.get("host_path"), d.get("cholesky_standalone"))
PY
tail -3 $O/bench.err

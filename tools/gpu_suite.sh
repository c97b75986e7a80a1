#!/bin/bash
# full GPU suite + the default bench exactly as the driver runs it + the small-N and sparse benches
export TMPDIR=/tmp
O=gpurun_out/${1:-r2o}
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3
( time timeout 900 python bench.py > $O/bench.json 2>> $O/bench.err ) 2> $O/bench_time.log
tail -4 $O/bench_time.log
for n in 2048 4096 8192; do timeout 300 python bench.py --n $n --d 8 --kind rbf --iso --steps 100 --warmup 10 --no-grid-leg --no-cpu-baseline > $O/bench_n$n.json 2>> $O/bench.err; done
timeout 600 python bench.py --sparse --steps 20 --warmup 3 > $O/bench_sparse.json 2>> $O/bench.err
python - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d.get("stage_ms"), {k: d.get(k) for k in ("cholesky_gflops", "iteration_frac_of_fp64_peak")}, d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("avg_launch_ms"), (d.get("grid") or {}).get("ms_per_step"), d.get("cpu_baseline", {}).get("value"))
PY
tail -3 $O/bench.err

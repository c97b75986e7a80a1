#!/bin/bash
# full GPU suite + the default bench exactly as the driver runs it (headline + c2 / grid / c5 legs)
export TMPDIR=/tmp
O=gpurun_out/${1:-r3a}
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -x ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
( time timeout 900 python bench.py > $O/bench.json 2>> $O/bench.err ) 2> $O/bench_time.log
tail -4 $O/bench_time.log
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print("C3", round(d["ms_per_step"], 3), d.get("stage_ms"), d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
for k in ("c2", "grid", "c5"):
    r = d.get(k) or {}
    print(k, r.get("error") or (round(r.get("ms_per_step", 0), 3), r.get("stage_ms"), (r.get("roofline") or {}).get("frac"), r.get("leg_wall_s"), (r.get("cpu_baseline") or {}).get("value"), r.get("parity") or r.get("parity_vs_golden")))
PY
tail -5 $O/bench.err

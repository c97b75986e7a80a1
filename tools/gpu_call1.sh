#!/bin/bash
# round 2, GPU call 1: full -m gpu suite, default bench, outer-panel-width sweep, kernel-trace stats, CPU baseline at full size
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 300 python - > $O/nbo.log 2>&1 <<'PY'
import os, json
from gpy_amd import _lib as L
res = {}
for n in (2048, 4096, 6144, 8192):
    for nbo in (128, 256, 512):
        os.environ["MI355GP_NBO"] = str(nbo)
        r = L.bench_factor(n, reps=5)
        res["%d/%d" % (n, nbo)] = r["potrf_ms"]
        print(n, nbo, "potrf %.3f trtri %.3f lauum %.3f" % (r["potrf_ms"], r["trtri_ms"], r["lauum_ms"]), flush=True)
os.environ.pop("MI355GP_NBO")
for n in (16384,):
    r = L.bench_factor(n, reps=3)
    print(n, "default", r)
PY
cat $O/nbo.log
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-grid-leg --no-parity-gate > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/stats -name "*stats*.csv" | head; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -25 $f
find $O/stats -name "*kernel_trace.csv" -size +30M -delete
timeout 600 python - > $O/cpu_full.log 2>&1 <<'PY'
import json, os, time, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.cpu_baseline("matern52", True, 32, 3072, 16384, full=True)
f = bench.cpu_baseline("matern52", True, 32, 3072, 16384, full=False)
json.dump({"full": r, "fit": f}, open("gpurun_out/r2a/cpu_full.json", "w"), indent=1)
print(json.dumps({"full": r, "fit": f}, indent=1))
PY
tail -40 $O/cpu_full.log

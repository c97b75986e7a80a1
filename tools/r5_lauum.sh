#!/bin/bash
export TMPDIR=/tmp
for n in 3072 4096; do
for t in 0 1 0 1; do
MI355GP_LAUUM_SPLIT=$t timeout 120 python - <<PY
import numpy as np, time
from gpy_amd import _lib as L
from gpy_amd.datasets import default_theta, synthetic
N, D = $n, 8
X, Y = synthetic(N, D, seed=0)
var, ls, noise = default_theta(D, False)
th = L.theta_vec(var, ls, False, D)
c = L.Context(0)
c.set_data(X, Y)
for _ in range(20): c.exact_inference("rbf", False, th, noise, want_alpha=False)
a0 = c.get_option("persist_aborts")
t0 = time.perf_counter()
for _ in range(300): info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False)
dt = (time.perf_counter() - t0) / 300
info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False, want_stage_ms=True)
print("N=$n lauum_split=$t: %.3f ms/step, aborts after warm-up %d, at the end %d, lml %.9f" % (1e3 * dt, a0, c.get_option("persist_aborts"), r["lml"]), {k: round(float(v), 3) for k, v in r["stage_ms"].items()})
c.close()
PY
done
done
timeout 200 python bench.py --sparse --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 ms_per_step %.3f' % d['ms_per_step'], d['stage_ms'])"
( time timeout 1200 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_baseline.py tests/test_gpu_persist_safety.py -m gpu -q --maxfail=5 -x ) 2>&1 | tail -5

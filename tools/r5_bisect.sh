#!/bin/bash
export TMPDIR=/tmp
for n in 2048 3072 4096 4608; do
for t in 0 1 0 1; do
MI355GP_PERSIST_TRI=$t timeout 120 python - <<PY
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
t0 = time.perf_counter()
for _ in range(300): info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False)
dt = (time.perf_counter() - t0) / 300
info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False, want_stage_ms=True)
print("N=$n tri=$t: %.3f ms/step, aborts %d lml %.6f" % (1e3 * dt, c.get_option("persist_aborts"), r["lml"]), {k: round(float(v), 3) for k, v in r["stage_ms"].items()})
c.close()
PY
done
done

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -x ) > $O/pytest.log 2>&1
grep -E "passed|failed|Error" $O/pytest.log | tail -5
python - <<'PY' > $O/kerr.log 2>&1
import sys, numpy as np
sys.path.insert(0, '.')
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic
import oracle.gp_oracle as O
for kind, ard, D in (("rbf", False, 8), ("matern52", True, 32), ("matern32", True, 5), ("rbf", True, 40)):
    X, Y = synthetic(1500, D, seed=1)
    X[7] = X[3]; X[11] = X[3] + 1e-7
    ls = np.linspace(0.5, 2.0, D) * np.sqrt(D / 8) if ard else np.sqrt(D) * 0.7
    th = L.theta_vec(1.3, ls, ard, D)
    K = L.kern_K(kind, ard, th, X)
    Ko = O.kern_K(kind, X, None, 1.3, ls, ard)
    Kx = L.kern_K(kind, ard, th, X[:700], X[300:])
    Kxo = O.kern_K(kind, X[:700], X[300:], 1.3, ls, ard)
    print(kind, ard, D, "max|dK|/var sym %.3g cross %.3g  dup entries %.17g %.17g" % (np.abs(K - Ko).max() / 1.3, np.abs(Kx - Kxo).max() / 1.3, K[7, 3], K[11, 3]), "sym", np.array_equal(K, K.T))
PY
cat $O/kerr.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-grid-leg --no-cpu-baseline --abi-only > $O/bench.json 2>> $O/bench.err
MI355GP_KERN_DIRECT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-grid-leg --no-cpu-baseline --abi-only > $O/bench_direct.json 2>> $O/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2m/bench*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d.get("stage_ms"), d.get("parity"))
PY
tail -3 $O/bench.err

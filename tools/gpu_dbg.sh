#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > t_dbg.py <<'PY'
import numpy as np, sys
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic, default_theta
n = int(sys.argv[1])
X, Y = synthetic(n, 8, seed=0)
var, ls, noise = default_theta(8, False)
c = L.Context(0); c.set_data(X, Y)
for k in range(3):
    info, r = c.exact_inference("rbf", False, L.theta_vec(var, ls, False, 8), noise, want_diag=True)
    print("OK", n, k, info, r["lml"], flush=True)
PY
for n in 300 2048 4096; do echo "== N=$n"; timeout 120 python t_dbg.py $n 2>&1 | tail -4; done
echo "== serialized 4096"; AMD_SERIALIZE_KERNEL=3 timeout 120 python t_dbg.py 4096 2>&1 | tail -4
echo "== logged 4096"; AMD_LOG_LEVEL=3 timeout 120 python t_dbg.py 4096 > gpurun_out/dbg.log 2>&1; grep -n "OK\|fault\|ShaderName" gpurun_out/dbg.log | tail -12 | cut -c1-200

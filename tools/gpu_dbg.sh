#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
cat > /tmp/one.py <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic, default_theta
n = 4096
X, Y = synthetic(n, 8, seed=0)
var, ls, noise = default_theta(8, False)
c = L.Context(0); c.set_data(X, Y)
th = L.theta_vec(var, ls, False, 8)
for _ in range(4):
    c.exact_inference("rbf", False, th, noise)
c.close()
PY
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace -d $O/tr -o run -- python /tmp/one.py > $O/tr.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
sed -i 's/"k_trsm128", /"k_trsm128", "k_next128", /' tools/chain_trace.py
python tools/chain_trace.py $f --show 70 > $O/chain.txt 2>&1
head -100 $O/chain.txt

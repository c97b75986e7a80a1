#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
cat > /tmp/one.py <<'PY'
import sys
sys.path.insert(0, '.')
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic, default_theta
n, D = 16384, 32
X, Y = synthetic(n, D, seed=0)
var, ls, noise = default_theta(D, True)
c = L.Context(0); c.set_data(X, Y)
th = L.theta_vec(var, ls, True, D)
for _ in range(3):
    c.exact_inference("matern52", True, th, noise)
c.close()
PY
rocprofv3 --output-format csv --kernel-trace -d $O/tr16k -o run -- python /tmp/one.py > $O/tr16k.log 2>&1
f=$(find $O/tr16k -name "*kernel_trace.csv" | head -1)
python tools/potrf_timeline.py $f > $O/potrf_timeline.txt 2>&1
head -40 $O/potrf_timeline.txt
rm -rf $O/tr16k

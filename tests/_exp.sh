python -m pytest tests/test_gpu_sparse.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import numpy as np, time, sys
sys.path.insert(0,'.')
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic, default_theta
N,M,D=200000,2048,16
X,Y=synthetic(N,D,seed=0)
Z=X[np.random.default_rng(1).permutation(N)[:M]].copy()
var,ls,noise=default_theta(D,False)
c=L.SparseContext(0); c.set_data(X,Y)
th=L.theta_vec(var,ls,False,D)
for it in range(4):
    t0=time.perf_counter(); info,r=c.vardtc("rbf",False,th,Z,noise,want_stage_ms=True); dt=time.perf_counter()-t0
    print(info, r["lml"], "wall %.1f ms"%(dt*1e3), {k:round(float(v),2) for k,v in r["stage_ms"].items()})
fl = 2.0*N*M*M/2 + 2.0*N*M*M
print("GEMM flops (psi2 lower + T): %.3e -> %.1f TF/s over total" % (fl, fl/(r["stage_ms"]["total"]*1e-3)/1e12))
PY

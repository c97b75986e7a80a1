import numpy as np, time, sys
sys.path.insert(0, "/root/repo")
from gpy_amd import _lib as L
from gpy_amd.datasets import default_theta, synthetic
for N in (4608, 5120, 12288):
    D = 8
    X, Y = synthetic(N, D, seed=0)
    var, ls, noise = default_theta(D, False)
    th = L.theta_vec(var, ls, False, D)
    c = L.Context(0)
    c.set_data(X, Y)
    for _ in range(5): c.exact_inference("rbf", False, th, noise, want_alpha=False)
    t0 = time.perf_counter()
    for _ in range(20): info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False)
    dt = (time.perf_counter() - t0) / 20
    info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False, want_stage_ms=True)
    print("N=%d: %.3f ms/step lml %.9f" % (N, 1e3 * dt, r["lml"]), {k: round(float(v), 3) for k, v in r["stage_ms"].items()})
    c.close()

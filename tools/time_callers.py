#!/usr/bin/env python3
"""GPU-box timing of the prediction-side callers and the linalg mirror at BASELINE configs[2]'s size (N=16384, D=32,
Matern-5/2 ARD): predict (diag / full covariance), predictive_gradients, pdinv.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpy_amd  # noqa: E402
from gpy_amd.datasets import default_theta, synthetic  # noqa: E402


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


def main():
    N, D, M = int(os.environ.get("N", 16384)), 32, int(os.environ.get("M", 4096))
    X, Y = synthetic(N, D, seed=0)
    var, ls, noise = default_theta(D, True)
    m = gpy_amd.GPRegression(X, Y, gpy_amd.Matern52(D, variance=var, lengthscale=ls, ARD=True), noise_var=noise)
    Xs = np.random.default_rng(1).standard_normal((M, D))
    out = {"N": N, "D": D, "M": M}
    out["predict_diag_ms"] = best(lambda: m.predict(Xs))
    out["predict_full_cov_ms"] = best(lambda: m.predict(Xs, full_cov=True))
    out["predictive_gradients_ms"] = best(lambda: m.predictive_gradients(Xs))
    out["predict_quantiles_ms"] = best(lambda: m.predict_quantiles(Xs))
    n2 = 8192
    A = np.asarray(m.kern.K(X[:n2])) + 0.1 * np.eye(n2)
    out["pdinv_n%d_ms_incl_pcie" % n2] = best(lambda: gpy_amd.linalg.pdinv(A), reps=2)
    out["jitchol_n%d_ms_incl_pcie" % n2] = best(lambda: gpy_amd.linalg.jitchol(A), reps=2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

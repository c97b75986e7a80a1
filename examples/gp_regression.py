#!/usr/bin/env python3
"""The reference's `GPy.examples.regression` style of use, on the MI355X backend:

    import gpy_amd as GPy
    m = GPy.models.GPRegression(X, Y, GPy.kern.Matern52(D, ARD=True))
    m.optimize()
    mu, var = m.predict(Xnew)

Needs an MI355X (no CPU fallback).  `python examples/gp_regression.py [N] [D]`"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpy_amd as GPy  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rng = np.random.default_rng(0)
    X = rng.uniform(-3, 3, (N, D))
    f = np.sin(X[:, 0]) * np.cos(0.5 * X[:, 1]) + 0.1 * X[:, 2 % D]
    Y = (f + 0.1 * rng.standard_normal(N))[:, None]
    Xs = rng.uniform(-3, 3, (512, D))
    fs = np.sin(Xs[:, 0]) * np.cos(0.5 * Xs[:, 1]) + 0.1 * Xs[:, 2 % D]

    m = GPy.models.GPRegression(X, Y, GPy.kern.Matern52(D, ARD=True), noise_var=0.5, normalizer=True)
    print("initial  log-likelihood %.3f" % m.log_likelihood())
    t0 = time.perf_counter()
    res = m.optimize(max_iters=60)
    dt = time.perf_counter() - t0
    print("optimised log-likelihood %.3f in %.2f s (%d objective + gradient evaluations, %.2f ms each)" % (
        m.log_likelihood(), dt, res.nfev, 1e3 * dt / max(res.nfev, 1)))
    print("lengthscales", np.round(m.kern.lengthscale.values, 3), "noise", np.round(m.likelihood.variance.values, 4))
    mu, var = m.predict(Xs)
    lo, hi = m.predict_quantiles(Xs)
    print("test RMSE %.4f, mean predictive sd %.4f, 95%% interval coverage %.3f" % (
        np.sqrt(np.mean((mu[:, 0] - fs) ** 2)), np.sqrt(var).mean(), np.mean((fs >= lo[:, 0]) & (fs <= hi[:, 0]))))
    dmu, dvar = m.predictive_gradients(Xs[:4])
    print("d mean / d x* at the first test point", np.round(dmu[0, :, 0], 4))

    ms = GPy.models.SparseGPRegression(X, Y, GPy.kern.RBF(D, ARD=True), num_inducing=64, noise_var=0.5, seed=1)
    t0 = time.perf_counter()
    ms.optimize(max_iters=40)
    mus, _ = ms.predict(Xs)
    print("sparse (M=64) log-likelihood %.3f after %.2f s, test RMSE %.4f" % (
        ms.log_likelihood(), time.perf_counter() - t0, np.sqrt(np.mean((mus[:, 0] - fs) ** 2))))


if __name__ == "__main__":
    main()

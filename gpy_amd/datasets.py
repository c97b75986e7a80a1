"""Synthetic regression data of the shape BASELINE.json's configurations name:
X ~ N(0,1)^{N x D}, Y = sin(x0) + 0.5 cos(2 x1) + 0.1 eps; default hyper-parameters sigma^2 = 1.3,
iso l = 0.7 sqrt(D) or ARD l_q = linspace(0.5, 2, D) sqrt(D/8), sigma_n^2 = 0.1."""
import numpy as np


def synthetic(N, D, seed=0, Dy=1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    f = np.sin(X[:, 0]) + (0.5 * np.cos(2 * X[:, 1]) if D >= 2 else 0.0)
    Y = (f + 0.1 * rng.standard_normal(N))[:, None]
    if Dy > 1:
        Y = np.hstack([Y + 0.3 * j * np.cos(X[:, :1] * (j + 1)) for j in range(Dy)])
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)


def default_theta(D, ARD):
    ls = np.linspace(0.5, 2.0, D) * np.sqrt(D / 8.0) if ARD else np.array([0.7 * np.sqrt(D)])
    return 1.3, ls, 0.1


def synthetic_Z(X, M, seed=0):
    """Inducing inputs Z = X[perm[:M]] (reference `models/sparse_gp_regression.py:41-43`), the draw the golden fixtures use."""
    rng = np.random.default_rng(seed + 77)
    return np.ascontiguousarray(X[rng.permutation(X.shape[0])[:M]].copy())

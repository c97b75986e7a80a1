"""CPU: the sparse (VarDTC) oracle against the golden vectors generated from the reference's own code
(oracle/make_golden_sparse.py) and, where /root/reference exists, against the reference run live.
Tolerances follow SURVEY.md 8(c): final quantities (LML rel 1e-9, theta / Z gradients rel 1e-6), not the
intermediate M x M matrices (Kmm + 1e-8 I is badly conditioned)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import sparse_oracle as S


def sparse_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "sparse_*.npz")))


def load_sparse_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["kind"] = str(d["kind"])
    d["ARD"] = bool(d["ARD"])
    d["variance"] = float(d["variance"])
    d["noise"] = float(d["noise"])
    return d


def check_sparse(res, g, tol_lml=1e-9, tol_g=1e-6):
    assert abs(res["lml"] - g["lml"]) <= tol_lml * abs(g["lml"])
    assert np.abs(res["dtheta"] - g["dtheta"]).max() <= tol_g * np.abs(g["dtheta"]).max()
    assert abs(res["dnoise"] - g["dnoise"]) <= tol_g * abs(g["dnoise"])
    assert np.abs(res["dZ"] - g["dZ"]).max() <= tol_g * np.abs(g["dZ"]).max()
    assert np.linalg.norm(res["woodbury_vector"] - g["woodbury_vector"]) <= 1e-5 * np.linalg.norm(g["woodbury_vector"])


@pytest.mark.parametrize("name", sparse_golden_names())
def test_sparse_oracle_matches_reference_golden(name):
    g = load_sparse_golden(name)
    ls = g["lengthscale"] if g["ARD"] else g["lengthscale"][:1]
    res = S.vardtc(g["kind"], g["X"], g["Z"], g["Y"], g["variance"], ls, g["ARD"], g["noise"])
    check_sparse(res, g)


def test_sparse_oracle_gradients_by_finite_differences():
    from oracle.gp_oracle import synthetic
    X, Y = synthetic(120, 2, seed=3)
    Z = S.synthetic_Z(X, 9, 0)
    var, ls, noise = 1.1, np.array([0.9, 1.7]), 0.2
    base = S.vardtc("matern52", X, Z, Y, var, ls, True, noise)
    eps = 1e-6
    f = lambda v, l, n, z: S.vardtc("matern52", X, z, Y, v, l, True, n)["lml"]   # noqa: E731
    assert abs((f(var + eps, ls, noise, Z) - f(var - eps, ls, noise, Z)) / (2 * eps) - base["dtheta"][0]) < 1e-4
    for q in range(2):
        d = np.zeros(2); d[q] = eps
        assert abs((f(var, ls + d, noise, Z) - f(var, ls - d, noise, Z)) / (2 * eps) - base["dtheta"][1 + q]) < 1e-4
    assert abs((f(var, ls, noise + eps, Z) - f(var, ls, noise - eps, Z)) / (2 * eps) - base["dnoise"]) < 1e-3
    Zp, Zm = Z.copy(), Z.copy()
    Zp[4, 1] += eps; Zm[4, 1] -= eps
    assert abs((f(var, ls, noise, Zp) - f(var, ls, noise, Zm)) / (2 * eps) - base["dZ"][4, 1]) < 1e-4


# ---- sum kernels, heteroscedastic noise, mean function (tests/golden/sparse2_*.npz, oracle/make_golden_sparse2.py) ----------
def sparse2_golden_names():
    # sparse2_*: oracle/make_golden_sparse2.py; sparse3_* (per-point noise with several output columns, D = 40):
    # oracle/make_golden_sparse3.py
    return sorted(os.path.splitext(os.path.basename(p))[0] for pat in ("sparse2_*.npz", "sparse3_*.npz")
                  for p in glob.glob(os.path.join(GOLDEN_DIR, pat)))


def load_sparse2_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    parts = []
    for i, kind in enumerate(d["kinds"]):
        ls = d["ls%d" % i]
        parts.append((str(kind), bool(d["ARDs"][i]), float(d["variances"][i]), ls if ls.size else None,
                      [int(v) for v in d["dims%d" % i]]) + ((int(d["terms"][i]),) if "terms" in d else ()))
    d["parts"] = parts
    d["R"] = d["Y"] - (d["X"] @ d["mean_w"] if d["mean_w"].size else 0.0)
    return d


def check_sparse2(res, g, tol_lml=1e-9, tol_g=1e-6):
    assert abs(res["lml"] - g["lml"]) <= tol_lml * abs(g["lml"])
    assert np.abs(res["dtheta"] - g["dtheta"]).max() <= tol_g * np.abs(g["dtheta"]).max()
    assert np.abs(np.atleast_1d(res["dnoise"]).ravel() - g["dnoise"]).max() <= tol_g * np.abs(g["dnoise"]).max()   # N x Dy, row-major
    assert np.abs(res["dZ"] - g["dZ"]).max() <= tol_g * np.abs(g["dZ"]).max()
    assert np.linalg.norm(res["woodbury_vector"] - g["woodbury_vector"]) <= 1e-5 * np.linalg.norm(g["woodbury_vector"])


@pytest.mark.parametrize("name", sparse2_golden_names())
def test_general_sparse_oracle_matches_reference_golden(name):
    g = load_sparse2_golden(name)
    res = S.vardtc_general(g["parts"], g["X"], g["Z"], g["R"], g["noise"])
    check_sparse2(res, g)
    assert np.abs(res["dL_dm"] - g["dL_dm"]).max() <= 1e-7 * np.abs(g["dL_dm"]).max()
    assert np.abs(res["dL_dKnm"][g["rows"]] - g["dL_dKnm_rows"]).max() <= 1e-6 * np.abs(g["dL_dKnm_rows"]).max()
    mu, var = S.sparse_predict(g["parts"], g["Z"], g["Xs"], res["woodbury_vector"], res["woodbury_inv"])
    _, cov = S.sparse_predict(g["parts"], g["Z"], g["Xs"], res["woodbury_vector"], res["woodbury_inv"], full_cov=True)
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-6 * np.abs(g["pred_mu"]).max()
    assert np.abs(var - g["pred_var"]).max() <= 1e-5 * np.abs(g["pred_var"]).max()
    assert np.abs(cov - g["pred_cov"]).max() <= 1e-5 * np.abs(g["pred_cov"]).max()

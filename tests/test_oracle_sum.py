"""CPU: the oracle's sum-kernel restatement (gp_oracle.sum_*) against golden vectors produced by the reference's own
`GPy.kern.Add` / `White` / `Bias` code (oracle/make_golden_sum.py)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import gp_oracle as O


def sum_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "sum_*.npz")))


def load_sum_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["noise"] = float(d["noise"])
    D = d["X"].shape[1]
    parts = []
    for kind, ARD, var, ls, dims in json.loads(str(d["specs"])):
        parts.append((kind, ARD, var, None if ls is None else np.array(ls),
                      np.arange(D) if dims is None else np.array(dims)))
    d["parts"] = parts
    return d


@pytest.mark.parametrize("name", sum_golden_names())
def test_sum_oracle_matches_reference_golden(name):
    g = load_sum_golden(name)
    r = O.sum_parameters_changed(g["parts"], g["X"], g["Y"], g["noise"])
    assert abs(r["lml"] - g["lml"]) <= 1e-11 * abs(g["lml"])
    assert np.abs(r["alpha"] - g["alpha"]).max() <= 1e-10 * np.abs(g["alpha"]).max()
    assert np.abs(r["dtheta"] - g["dtheta"]).max() <= 1e-9 * np.abs(g["dtheta"]).max()
    assert abs(r["dL_dnoise"] - g["dnoise"]) <= 1e-10 * abs(g["dnoise"])
    assert np.abs(r["K"][0] - g["K_row0"]).max() <= 1e-14
    mu, var, cov = O.sum_predict(g["parts"], g["X"], g["Xs"], r["L"], r["alpha"])
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-9 and np.abs(var - g["pred_var"]).max() <= 1e-9
    assert np.abs(cov - g["pred_cov"]).max() <= 1e-9

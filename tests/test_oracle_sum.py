"""CPU: the oracle's sum / product kernel restatement (gp_oracle.sum_*) against golden vectors produced by the reference's
own `GPy.kern.Add` / `Prod` / `White` / `Bias` code (oracle/make_golden_sum.py, oracle/make_golden_prod.py)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import gp_oracle as O


def sum_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for pat in ("sum_*.npz", "prod_*.npz")
                  for p in glob.glob(os.path.join(GOLDEN_DIR, pat)))


def load_sum_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["noise"] = float(d["noise"])
    D = d["X"].shape[1]
    parts = []
    for spec in json.loads(str(d["specs"])):          # prod_* fixtures carry a 6th entry: the term id
        kind, ARD, var, ls, dims = spec[:5]
        parts.append((kind, ARD, var, None if ls is None else np.array(ls),
                      np.arange(D) if dims is None else np.array(dims), spec[5] if len(spec) > 5 else 0))
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

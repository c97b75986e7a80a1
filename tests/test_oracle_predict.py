"""CPU: the oracle's restatement of the prediction-side callers of the exact path (core/gp.py: predict_quantiles,
log_predictive_density, predictive_gradients, posterior_covariance_between_points) against golden vectors produced by the
reference's own kernel / posterior / likelihood objects (oracle/make_golden_predict.py)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import gp_oracle as O


def predict_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "predict_*.npz")))


def load_predict_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["kind"], d["ARD"], d["variance"], d["noise"] = str(d["kind"]), bool(d["ARD"]), float(d["variance"]), float(d["noise"])
    d["ls"] = d["lengthscale"] if d["ARD"] else d["lengthscale"][:1]
    return d


def test_there_are_prediction_goldens():
    assert len(predict_golden_names()) >= 3


@pytest.mark.parametrize("name", predict_golden_names())
def test_prediction_side_oracle_matches_reference_golden(name):
    g = load_predict_golden(name)
    K = O.kern_K(g["kind"], g["X"], None, g["variance"], g["ls"], g["ARD"])
    r = O.exact_inference(K, g["Y"], g["noise"])
    assert abs(r["lml"] - g["lml"]) <= 1e-11 * abs(g["lml"])
    mu, var = O.predict(g["kind"], g["X"], g["Xs"], r["L"], r["alpha"], g["variance"], g["ls"], g["ARD"])
    assert np.abs(mu - g["mu"]).max() <= 1e-9 * np.abs(g["mu"]).max()
    assert np.abs(var - g["var"]).max() <= 1e-9 * np.abs(g["var"]).max()
    q = O.predictive_quantiles(mu, var, g["noise"], (2.5, 50.0, 97.5))
    assert np.abs(np.stack(q) - g["quantiles"]).max() <= 1e-9 * np.abs(g["quantiles"]).max()
    lpd = O.log_predictive_density(g["ys"], mu, var, g["noise"])
    assert np.abs(lpd - g["lpd"]).max() <= 1e-9 * np.abs(g["lpd"]).max()
    mj, vj = O.predictive_gradients(g["kind"], g["X"], g["Xs"], r["alpha"], r["Wi"], g["variance"], g["ls"], g["ARD"])
    assert np.abs(mj - g["mean_jac"]).max() <= 1e-8 * np.abs(g["mean_jac"]).max()
    assert np.abs(vj - g["var_jac"]).max() <= 1e-7 * np.abs(g["var_jac"]).max()
    # finite differences of the oracle's own prediction pin the sign / layout conventions independently of the reference
    h = 1e-6
    m, qd = 3, g["Xs"].shape[1] - 1
    Xp, Xm = g["Xs"].copy(), g["Xs"].copy()
    Xp[m, qd] += h
    Xm[m, qd] -= h
    mp_, vp_ = O.predict(g["kind"], g["X"], Xp, r["L"], r["alpha"], g["variance"], g["ls"], g["ARD"])
    mm_, vm_ = O.predict(g["kind"], g["X"], Xm, r["L"], r["alpha"], g["variance"], g["ls"], g["ARD"])
    assert np.allclose((mp_[m] - mm_[m]) / (2 * h), g["mean_jac"][m, qd, :], rtol=1e-5, atol=1e-7)
    assert np.allclose((vp_[m] - vm_[m]) / (2 * h), g["var_jac"][m, qd], rtol=1e-4, atol=1e-7)

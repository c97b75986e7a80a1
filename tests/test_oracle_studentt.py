"""CPU: the oracle's Student-t process restatement against golden vectors from the reference's own
ExactStudentTInference (oracle/make_golden_studentt.py)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import gp_oracle as O


def studentt_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "studentt_*.npz")))


def load_studentt_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["kind"], d["ARD"], d["variance"], d["nu"] = str(d["kind"]), bool(d["ARD"]), float(d["variance"]), float(d["nu"])
    d["ls"] = d["lengthscale"] if d["ARD"] else d["lengthscale"][:1]
    return d


@pytest.mark.parametrize("name", studentt_golden_names())
def test_studentt_oracle_matches_reference_golden(name):
    g = load_studentt_golden(name)
    K = O.kern_K(g["kind"], g["X"], None, g["variance"], g["ls"], g["ARD"])
    r = O.studentt_inference(K, g["Y"], g["nu"])
    assert abs(r["lml"] - g["lml"]) <= 1e-11 * abs(g["lml"])
    assert np.abs(r["alpha"] - g["alpha"]).max() <= 1e-8 * np.abs(g["alpha"]).max()
    assert abs(r["dL_dnu"] - g["dL_dnu"]) <= 1e-9 * abs(g["dL_dnu"])
    assert np.abs(r["dL_dm"] - g["dL_dm"]).max() <= 1e-8 * np.abs(g["dL_dm"]).max()
    dv, dl = O.update_gradients_full(g["kind"], r["dL_dK"], g["X"], None, g["variance"], g["ls"], g["ARD"])
    got = np.concatenate([[dv], np.atleast_1d(dl)])
    assert np.abs(got - g["dtheta"]).max() <= 1e-7 * np.abs(g["dtheta"]).max()
    # StudentTPosterior._raw_predict (posterior.py:338-349): Gaussian prediction, variance scaled by (nu+beta-2)/(nu+N-2)
    mu, var = O.predict(g["kind"], g["X"], g["Xs"], r["L"], r["alpha"], g["variance"], g["ls"], g["ARD"])
    _, cov = O.predict(g["kind"], g["X"], g["Xs"], r["L"], r["alpha"], g["variance"], g["ls"], g["ARD"], full_cov=True)
    N = g["X"].shape[0]
    scale = (g["nu"] + float(np.sum(r["alpha"] * (K @ r["alpha"]))) - 2.0) / (g["nu"] + N - 2.0)
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-6 * np.abs(g["pred_mu"]).max()
    assert np.abs(scale * var - g["pred_var"]).max() <= 1e-6 * np.abs(g["pred_var"]).max()
    assert np.abs(scale * cov - g["pred_cov"]).max() <= 1e-6 * np.abs(g["pred_cov"]).max()
    assert np.abs(r["dL_dK"][g["rows"]] - g["dL_dK_rows"]).max() <= 1e-6 * np.abs(g["dL_dK_rows"]).max()

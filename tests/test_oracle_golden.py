"""CPU: the oracle (oracle/gp_oracle.py) reproduces every golden vector generated from the reference's own code
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP path with both."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import gp_oracle as O


def _ls(g):
    return g["lengthscale"] if g["ARD"] else g["lengthscale"][:1]


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name, oracle_native_built):
    g = load_golden(name)
    X, Y = g["X"], g["Y"]
    noise = g["noise"] if g["noise"].size > 1 else float(g["noise"][0])
    r = O.parameters_changed(g["kind"], X, Y, g["variance"], _ls(g), g["ARD"], noise)
    rows = g["rows"]
    assert abs(r["lml"] - g["lml"]) <= 1e-12 * max(1.0, abs(g["lml"]))
    np.testing.assert_allclose(r["alpha"], g["alpha"], rtol=0, atol=1e-12 * np.abs(g["alpha"]).max())
    np.testing.assert_allclose(r["K"][rows], g["K_rows"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(r["L"][rows], g["L_rows"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r["dL_dK"][rows], g["dL_dK_rows"], rtol=0, atol=1e-12 * np.abs(g["dL_dK_rows"]).max())
    np.testing.assert_allclose(r["diag_dL_dK"], g["diag_dL_dK"], rtol=0, atol=1e-12 * np.abs(g["diag_dL_dK"]).max())
    assert abs(r["dvar"] - g["dvar"][0]) <= 1e-11 * abs(g["dvar"][0])
    # the C helper sums the Q*N*M loop in a different order than NumPy's pairwise sum: 1e-10 relative
    np.testing.assert_allclose(r["dlen"], g["dlen"], rtol=1e-10)
    if g["noise"].size == 1:
        assert abs(r["dL_dnoise"] - g["dnoise"][0]) <= 1e-11 * abs(g["dnoise"][0])
    np.testing.assert_allclose(O.kern_Kdiag(X, g["variance"]), g["Kdiag"], rtol=0, atol=0)
    # generic update_gradients_full (non-symmetric dL_dK, rectangular K)
    dv, dl = O.update_gradients_full(g["kind"], g["A"], X, g["X2"], g["variance"], _ls(g), g["ARD"])
    assert abs(dv - g["dvar_A"][0]) <= 1e-11 * max(1.0, abs(g["dvar_A"][0]))
    np.testing.assert_allclose(dl, g["dlen_A"], rtol=1e-10, atol=1e-12)
    Kx = O.kern_K(g["kind"], X, g["X2"], g["variance"], _ls(g), g["ARD"])
    np.testing.assert_allclose(Kx[rows], g["K_X_X2_rows"], rtol=0, atol=1e-15)
    if "pred_mu" in g:
        mu, var = O.predict(g["kind"], X, g["Xs"], r["L"], r["alpha"], g["variance"], _ls(g), g["ARD"])
        np.testing.assert_allclose(mu, g["pred_mu"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(var, g["pred_var"], rtol=0, atol=1e-12)
        _, cov = O.predict(g["kind"], X, g["Xs"], r["L"], r["alpha"], g["variance"], _ls(g), g["ARD"], full_cov=True)
        np.testing.assert_allclose(cov, g["pred_cov"], rtol=0, atol=1e-12)


def test_jitchol_ladder_semantics():
    """reference GPy/testing/test_linalg.py:20-37: a rank-deficient matrix factors only with the ladder."""
    A = np.ones((4, 4))
    L = O.jitchol(A)
    assert np.allclose(L @ L.T, A, atol=1e-4)
    with pytest.raises(np.linalg.LinAlgError):
        O.jitchol(np.array([[1.0, 2.0], [2.0, -1.0]]))      # non-positive diagonal
    with pytest.raises(np.linalg.LinAlgError):
        O.jitchol(np.array([[1.0, 5.0], [5.0, 1.0]]))       # indefinite even with jitter


def test_gradients_match_finite_differences():
    """the style of the reference's own checks (checkgrad, GPy/testing/test_model.py:790-898)."""
    X, Y = O.synthetic(120, 3, seed=5)
    for kind in O.KINDS:
        var, ls, noise = 1.1, np.array([0.7, 1.3, 2.0]), 0.2
        r = O.parameters_changed(kind, X, Y, var, ls, True, noise)
        g = np.concatenate([[r["dvar"]], r["dlen"], [r["dL_dnoise"]]])
        x0 = np.concatenate([[var], ls, [noise]])
        fd = np.zeros_like(x0)
        for i in range(x0.size):
            h = 1e-6 * x0[i]
            xp, xm = x0.copy(), x0.copy()
            xp[i] += h
            xm[i] -= h
            fp = O.parameters_changed(kind, X, Y, xp[0], xp[1:4], True, xp[4])["lml"]
            fm = O.parameters_changed(kind, X, Y, xm[0], xm[1:4], True, xm[4])["lml"]
            fd[i] = (fp - fm) / (2 * h)
        np.testing.assert_allclose(g, fd, rtol=2e-5, atol=1e-6)

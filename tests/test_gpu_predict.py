"""GPU (-m gpu): the prediction-side callers of the exact path (core/gp.py: predict_quantiles, log_predictive_density,
predictive_gradients, posterior_covariance_between_points, posterior_samples) through the host classes and the C-ABI
(`mi355gp_predictive_gradients_sum`), against golden vectors from the reference's own kernel / posterior / likelihood objects."""
import numpy as np
import pytest

import gpy_amd
from gpy_amd import _lib as L
from oracle import gp_oracle as O
from test_oracle_predict import load_predict_golden, predict_golden_names

pytestmark = pytest.mark.gpu
KCLS = {"rbf": gpy_amd.RBF, "matern52": gpy_amd.Matern52, "matern32": gpy_amd.Matern32}


def _model(g, **kw):
    D = g["X"].shape[1]
    ls = g["lengthscale"] if g["ARD"] else float(g["lengthscale"][0])
    k = KCLS[g["kind"]](D, variance=g["variance"], lengthscale=ls, ARD=g["ARD"])
    return gpy_amd.GPRegression(g["X"], g["Y"], k, noise_var=g["noise"], **kw)


@pytest.mark.parametrize("name", predict_golden_names())
def test_prediction_callers_match_reference_golden(name):
    g = load_predict_golden(name)
    m = _model(g)
    assert abs(m.log_likelihood() - g["lml"]) <= 1e-10 * abs(g["lml"])
    q = m.predict_quantiles(g["Xs"], quantiles=(2.5, 50.0, 97.5))
    assert np.abs(np.stack(q) - g["quantiles"]).max() <= 1e-9 * np.abs(g["quantiles"]).max()
    lpd = m.log_predictive_density(g["Xs"], g["ys"])
    assert np.abs(lpd - g["lpd"]).max() <= 1e-9 * np.abs(g["lpd"]).max()
    mj, vj = m.predictive_gradients(g["Xs"])
    assert mj.shape == g["mean_jac"].shape and vj.shape == g["var_jac"].shape
    assert np.abs(mj - g["mean_jac"]).max() <= 1e-8 * np.abs(g["mean_jac"]).max()
    assert np.abs(vj - g["var_jac"]).max() <= 1e-7 * np.abs(g["var_jac"]).max()
    cov = m.posterior_covariance_between_points(g["Xs"][:9], g["Xs"][:9], include_likelihood=False)
    assert np.abs(cov - g["cov"]).max() <= 1e-9
    covl = m.posterior_covariance_between_points(g["Xs"][:9], g["Xs"][:9])
    assert np.abs(covl - g["cov_lik"]).max() <= 1e-9
    # the host composition (what product kernels take) agrees with the device reduction
    post = m.posterior
    st, post._state = post._state, None
    try:
        with pytest.raises(RuntimeError):
            post.predictive_gradients(m.kern, g["Xs"])
    finally:
        post._state = st


def test_predictive_gradients_of_sum_and_product_kernels_and_normaliser():
    X, Y = O.synthetic(500, 4, seed=5)
    rng = np.random.default_rng(9)
    Xs = rng.standard_normal((21, 4))
    k = gpy_amd.RBF(2, variance=0.9, lengthscale=1.1, active_dims=[0, 2]) + \
        gpy_amd.Matern52(3, variance=0.7, lengthscale=[0.8, 1.3, 2.0], ARD=True, active_dims=[1, 2, 3]) + gpy_amd.White(4, 0.02) + \
        gpy_amd.Bias(4, 0.3)
    m = gpy_amd.GPRegression(X, Y, k, noise_var=0.05, normalizer=True)
    mj, vj = m.predictive_gradients(Xs)
    assert mj.shape == (21, 4, 1) and vj.shape == (21, 4)
    # central differences of the model's own (device) prediction
    h = 1e-6
    for mi, q in ((0, 0), (5, 2), (11, 3), (20, 1)):
        Xp, Xm = Xs.copy(), Xs.copy()
        Xp[mi, q] += h
        Xm[mi, q] -= h
        mup, vp = m.predict_noiseless(Xp)
        mum, vm = m.predict_noiseless(Xm)
        assert np.allclose((mup[mi] - mum[mi]) / (2 * h), mj[mi, q], rtol=2e-5, atol=1e-7)
        assert np.allclose((vp[mi] - vm[mi]) / (2 * h), vj[mi, q], rtol=2e-4, atol=1e-7)
    # a product kernel takes the host composition (kern.gradients_X + fetched woodbury_inv): same check
    kp = gpy_amd.RBF(2, variance=0.9, lengthscale=1.1, active_dims=[0, 1]) * gpy_amd.Matern32(2, variance=1.2, lengthscale=0.9,
                                                                                               active_dims=[2, 3])
    mp = gpy_amd.GPRegression(X, Y, kp, noise_var=0.05)
    mj2, vj2 = mp.predictive_gradients(Xs)
    for mi, q in ((1, 0), (7, 3)):
        Xp, Xm = Xs.copy(), Xs.copy()
        Xp[mi, q] += h
        Xm[mi, q] -= h
        mup, vp = mp.predict_noiseless(Xp)
        mum, vm = mp.predict_noiseless(Xm)
        assert np.allclose((mup[mi] - mum[mi]) / (2 * h), mj2[mi, q], rtol=2e-5, atol=1e-7)
        assert np.allclose((vp[mi] - vm[mi]) / (2 * h), vj2[mi, q], rtol=2e-4, atol=1e-7)


def test_posterior_samples_have_the_predicted_moments():
    X, Y = O.synthetic(300, 2, seed=8)
    m = gpy_amd.GPRegression(X, Y, gpy_amd.RBF(2, variance=1.0, lengthscale=1.2), noise_var=0.1)
    Xs = np.random.default_rng(3).standard_normal((6, 2))
    np.random.seed(0)
    f = m.posterior_samples_f(Xs, size=4000)
    assert f.shape == (6, 1, 4000)
    mu, cov = m.predict_noiseless(Xs, full_cov=True)
    assert np.abs(f.mean(-1) - mu).max() <= 5 * np.sqrt(np.diag(cov).max() / 4000) + 1e-3
    y = m.posterior_samples(Xs, size=4000)
    assert y.shape == (6, 1, 4000) and y.var(-1).mean() > f.var(-1).mean()
    m.set_Y(Y * 2.0)
    mu2, _ = m.predict_noiseless(Xs)
    assert np.allclose(mu2, 2.0 * mu, rtol=1e-8, atol=1e-10)


def test_predictive_gradients_entry_point_at_a_larger_size():
    """N = 3000 (24 tiles): the X^T (X Kx) products run on several tile rows; checked against the oracle."""
    X, Y = O.synthetic(3000, 6, seed=2)
    var, ls, noise = O.default_theta(6, True)
    th = L.theta_vec(var, ls, True, 6)
    Xs = np.random.default_rng(1).standard_normal((150, 6))
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        info, r = c.exact_inference("matern52", True, th, noise)
        assert info == 0
        dmu, dvar = c.predictive_gradients([("matern52", True, th, None)], Xs)
    finally:
        c.close()
    ref = O.exact_inference(O.kern_K("matern52", X, None, var, ls, True), Y, noise)
    mj, vj = O.predictive_gradients("matern52", X, Xs, ref["alpha"], ref["Wi"], var, ls, True)
    assert np.abs(dmu - mj).max() <= 1e-8 * np.abs(mj).max()
    assert np.abs(dvar - vj).max() <= 1e-7 * np.abs(vj).max()


def test_predictive_gradients_with_more_than_32_input_dimensions():
    """D = 40: the n-reductions of `GP.predictive_gradients` (core/gp.py:418-474) in two groups of input dimensions."""
    D = 40
    X, Y = O.synthetic(900, D, seed=6)
    var, ls, noise = O.default_theta(D, True)
    th = L.theta_vec(var, ls, True, D)
    Xs = np.random.default_rng(2).standard_normal((70, D))
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        info, r = c.exact_inference("rbf", True, th, noise)
        assert info == 0
        dmu, dvar = c.predictive_gradients([("rbf", True, th, None)], Xs)
    finally:
        c.close()
    ref = O.exact_inference(O.kern_K("rbf", X, None, var, ls, True), Y, noise)
    mj, vj = O.predictive_gradients("rbf", X, Xs, ref["alpha"], ref["Wi"], var, ls, True)
    assert np.abs(dmu - mj).max() <= 1e-8 * np.abs(mj).max()
    assert np.abs(dvar - vj).max() <= 1e-7 * np.abs(vj).max()


def _sparse_names():
    import glob
    import os
    from conftest import GOLDEN_DIR
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "predsparse_*.npz")))


@pytest.mark.parametrize("name", _sparse_names())
def test_sparse_model_prediction_callers_match_reference_golden(name):
    """SparseGP inherits the same callers (core/sparse_gp.py: `_predictive_variable` = Z): golden vectors from the reference's
    VarDTC posterior."""
    import os
    from conftest import GOLDEN_DIR
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    kind, ARD = str(g["kind"]), bool(g["ARD"])
    D = g["X"].shape[1]
    ls = g["lengthscale"] if ARD else float(g["lengthscale"][0])
    k = KCLS[kind](D, variance=float(g["variance"]), lengthscale=ls, ARD=ARD)
    m = gpy_amd.SparseGPRegression(g["X"], g["Y"], kernel=k, Z=g["Z"], noise_var=float(g["noise"]))
    assert abs(m.log_likelihood() - float(g["lml"])) <= 1e-9 * abs(float(g["lml"]))
    q = m.predict_quantiles(g["Xs"])
    assert np.abs(np.stack(q) - g["quantiles"]).max() <= 1e-7 * np.abs(g["quantiles"]).max()
    lpd = m.log_predictive_density(g["Xs"], g["ys"])
    assert np.abs(lpd - g["lpd"]).max() <= 1e-7 * np.abs(g["lpd"]).max()
    mj, vj = m.predictive_gradients(g["Xs"])
    assert np.abs(mj - g["mean_jac"]).max() <= 1e-6 * np.abs(g["mean_jac"]).max()
    assert np.abs(vj - g["var_jac"]).max() <= 1e-6 * np.abs(g["var_jac"]).max()
    # posterior covariance between points: consistent with the model's own full-covariance prediction
    cov = m.posterior_covariance_between_points(g["Xs"][:8], g["Xs"][:8], include_likelihood=False)
    _, full = m.predict_noiseless(g["Xs"][:8], full_cov=True)
    assert np.abs(cov - full).max() <= 1e-8
    f = m.posterior_samples_f(g["Xs"][:5], size=3)
    assert f.shape == (5, m.output_dim, 3)


def test_heteroscedastic_regression_model_matches_oracle():
    """GPHeteroscedasticRegression (models/gp_heteroscedastic_regression.py): per-point noise in, N noise gradients out."""
    X, Y = O.synthetic(300, 3, seed=6)
    k = gpy_amd.Matern32(3, variance=1.1, lengthscale=[0.9, 1.4, 2.0], ARD=True)
    m = gpy_amd.GPHeteroscedasticRegression(X, Y, k)
    noise = np.random.default_rng(2).uniform(0.03, 0.3, 300)
    m.likelihood.variance[:] = noise
    m.parameters_changed()
    K = O.kern_K("matern32", X, None, 1.1, np.array([0.9, 1.4, 2.0]), True)
    ref = O.exact_inference(K, Y, noise)
    assert abs(m.log_likelihood() - ref["lml"]) <= 1e-10 * abs(ref["lml"])
    assert np.abs(m.likelihood.variance.gradient - np.diag(ref["dL_dK"])).max() <= 1e-8 * np.abs(np.diag(ref["dL_dK"])).max()
    dv, dl = O.update_gradients_full("matern32", ref["dL_dK"], X, None, 1.1, np.array([0.9, 1.4, 2.0]), True)
    assert np.abs(m.kern.gradient - np.concatenate([[dv], dl])).max() <= 1e-8 * np.abs(dl).max()
    mu, var = m.predict(X[:5], Y_metadata={"output_index": np.arange(5)[:, None]})
    mu0, var0 = m.predict_noiseless(X[:5])
    assert np.allclose(var - var0, noise[:5, None], rtol=1e-9, atol=1e-12) and np.allclose(mu, mu0)

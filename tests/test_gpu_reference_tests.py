"""GPU (-m gpu): the reference's own known-answer TESTS of this path, restated against the device classes:
  (ii)  `testing/test_model.py:83-105`  test_raw_predict: predict_noiseless == the explicit pinv formula to 7 decimals
  (iii) `testing/test_gpy_kernels_state_space.py:55-130`: GPRegression with a Matern-3/2 / Matern-5/2 kernel on 1-D inputs has
        the same log marginal likelihood and predictions as the equivalent linear-Gaussian state-space model (here an
        independent Kalman filter / RTS-free prediction written from the SDE form, not the reference's state_space_model.py)
  (iv)  `testing/test_model.py:1250-1283`: a product kernel over a full grid == the Kronecker-structured evaluation"""
import numpy as np
import pytest
from scipy.linalg import expm

import gpy_amd

pytestmark = pytest.mark.gpu


def test_raw_predict_equals_the_pinv_formula():
    rng = np.random.RandomState(0)
    N, N_new, D = 100, 20, 1
    X = np.linspace(0, 10, N)[:, None]
    X_new = np.random.RandomState(1).uniform(0, 10, (N_new, 1))
    Y = np.sin(X) + rng.randn(N, D) * 0.05
    k = gpy_amd.RBF(1, variance=1.7, lengthscale=0.9)
    m = gpy_amd.GPRegression(X, Y, kernel=k, noise_var=0.5)
    Kinv = np.linalg.pinv(np.asarray(k.K(X)) + np.eye(N) * 0.5)
    K_hat = np.asarray(k.K(X_new)) - np.asarray(k.K(X_new, X)).dot(Kinv).dot(np.asarray(k.K(X, X_new)))
    mu_hat = np.asarray(k.K(X_new, X)).dot(Kinv).dot(m.Y_normalized)
    mu, covar = m.predict_noiseless(X_new, full_cov=True)
    assert mu.shape == (N_new, D) and covar.shape == (N_new, N_new)
    np.testing.assert_almost_equal(K_hat, covar)
    np.testing.assert_almost_equal(mu_hat, mu)
    mu, var = m.predict_noiseless(X_new)
    assert mu.shape == (N_new, D) and var.shape == (N_new, 1)
    np.testing.assert_almost_equal(np.diag(K_hat)[:, None], var)
    np.testing.assert_almost_equal(mu_hat, mu)


def _sde(kind, variance, ell):
    """stationary SDE form: dx = F x dt + noise, f = H x, stationary covariance Pinf"""
    if kind == "matern32":
        lam = np.sqrt(3.0) / ell
        F = np.array([[0.0, 1.0], [-lam ** 2, -2 * lam]])
        Pinf = np.diag([variance, lam ** 2 * variance])
    else:
        lam = np.sqrt(5.0) / ell
        F = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3 * lam ** 2, -3 * lam]])
        kappa = 5.0 / 3.0 * variance / ell ** 2
        Pinf = np.array([[variance, 0.0, -kappa], [0.0, kappa, 0.0], [-kappa, 0.0, 25.0 * variance / ell ** 4]])
    H = np.zeros((1, F.shape[0]))
    H[0, 0] = 1.0
    return F, Pinf, H


def _kalman_lml_and_filter(kind, variance, ell, noise, t, y, t_new):
    """log p(y) by the Kalman filter over the sorted times; predictive mean / variance of f at times AFTER the data
    (pure forward prediction, so no smoother is needed)."""
    F, Pinf, H = _sde(kind, variance, ell)
    m = np.zeros(F.shape[0])
    P = Pinf.copy()
    lml, tp = 0.0, t[0]
    for ti, yi in zip(t, y):
        A = expm(F * (ti - tp))
        m, P = A @ m, A @ P @ A.T + (Pinf - A @ Pinf @ A.T)
        s = (H @ P @ H.T).item() + noise
        v = yi - (H @ m).item()
        lml += -0.5 * (np.log(2 * np.pi * s) + v * v / s)
        Kg = (P @ H.T / s).ravel()
        m, P = m + Kg * v, P - np.outer(Kg, Kg) * s
        tp = ti
    mus, vs = [], []
    for tn in t_new:
        A = expm(F * (tn - tp))
        mn, Pn = A @ m, A @ P @ A.T + (Pinf - A @ Pinf @ A.T)
        mus.append((H @ mn).item())
        vs.append((H @ Pn @ H.T).item())
    return lml, np.array(mus), np.array(vs)


@pytest.mark.parametrize("kind,cls", [("matern32", gpy_amd.Matern32), ("matern52", gpy_amd.Matern52)])
def test_matern_regression_equals_the_state_space_model(kind, cls):
    rng = np.random.default_rng(4)
    N = 600
    t = np.sort(rng.uniform(0.0, 30.0, N))
    y = np.sin(t) + 0.3 * np.cos(3.1 * t) + 0.1 * rng.standard_normal(N)
    variance, ell, noise = 1.4, 0.8, 0.02
    t_new = 30.0 + np.array([0.05, 0.4, 1.0, 2.5])
    # the exact path factorises K + (noise + 1e-8) I (exact_gaussian_inference.py:55-56): the filter gets the same noise
    lml, mu_ss, var_ss = _kalman_lml_and_filter(kind, variance, ell, noise + 1e-8, t, y, t_new)
    m = gpy_amd.GPRegression(t[:, None], y[:, None], cls(1, variance=variance, lengthscale=ell), noise_var=noise)
    assert abs(m.log_likelihood() - lml) <= 1e-7 * abs(lml)
    mu, var = m.predict_noiseless(t_new[:, None])
    np.testing.assert_allclose(mu.ravel(), mu_ss, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var.ravel(), var_ss, rtol=1e-6, atol=1e-8)


def test_product_kernel_on_a_grid_equals_the_kronecker_evaluation():
    """K = K1 (x) K2 on a full grid: LML and alpha from the eigendecompositions of the two small factors
    (the algebra of GPKroneckerGaussianRegression, models/gp_kronecker_gaussian_regression.py:75-100)."""
    rng = np.random.default_rng(6)
    x1 = np.sort(rng.uniform(0, 4, 23))
    x2 = np.sort(rng.uniform(0, 3, 17))
    G1, G2 = np.meshgrid(x1, x2, indexing="ij")
    X = np.stack([G1.ravel(), G2.ravel()], 1)
    Y = (np.sin(G1) * np.cos(2 * G2)).ravel()[:, None] + 0.05 * rng.standard_normal((X.shape[0], 1))
    k = gpy_amd.RBF(1, variance=1.3, lengthscale=0.7, active_dims=[0]) * gpy_amd.Matern32(1, variance=0.9, lengthscale=1.1,
                                                                                           active_dims=[1])
    noise = 0.07
    m = gpy_amd.GPRegression(X, Y, k, noise_var=noise)
    K1 = np.asarray(gpy_amd.RBF(1, variance=1.3, lengthscale=0.7).K(x1[:, None]))
    K2 = np.asarray(gpy_amd.Matern32(1, variance=0.9, lengthscale=1.1).K(x2[:, None]))
    S1, U1 = np.linalg.eigh(K1)
    S2, U2 = np.linalg.eigh(K2)
    W = np.kron(S1, S2) + noise + 1e-8
    Yr = Y.reshape(len(x1), len(x2))
    Yt = U1.T @ Yr @ U2
    alpha = (U1 @ (Yt / W.reshape(len(x1), len(x2))) @ U2.T).reshape(-1, 1)
    lml = -0.5 * X.shape[0] * np.log(2 * np.pi) - 0.5 * np.sum(np.log(W)) - 0.5 * float(np.sum(alpha * Y))
    assert abs(m.log_likelihood() - lml) <= 1e-9 * abs(lml)
    assert np.allclose(m.posterior.woodbury_vector, alpha, rtol=1e-7, atol=1e-9)


def test_checkgrad_like_the_reference_tests():
    """`assert m.checkgrad()` on the model families of this path (testing/test_model.py:790-898, test_kernel.py:57-70)."""
    rng = np.random.default_rng(8)
    X = rng.standard_normal((150, 3))
    Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((150, 1))
    np.random.seed(1)
    for k in (gpy_amd.RBF(3, ARD=True, lengthscale=[0.9, 1.3, 2.0]), gpy_amd.Matern52(3) + gpy_amd.White(3, 0.1),
              gpy_amd.RBF(2, active_dims=[0, 1]) * gpy_amd.Matern32(1, active_dims=[2]), gpy_amd.Exponential(3, lengthscale=1.5)):
        m = gpy_amd.GPRegression(X, Y, k, noise_var=0.2)
        assert m.checkgrad(), k
    ms = gpy_amd.SparseGPRegression(X, Y, kernel=gpy_amd.RBF(3, lengthscale=1.2), num_inducing=15, noise_var=0.2, seed=0)
    assert ms.checkgrad()
    mh = gpy_amd.GPHeteroscedasticRegression(X, Y, gpy_amd.RBF(3))
    assert mh.checkgrad()

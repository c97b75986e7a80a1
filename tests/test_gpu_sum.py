"""GPU (-m gpu): sum kernels (GPy.kern.Add of stationary + White + Bias parts, with active_dims) through the fused C-ABI
call `mi355gp_exact_inference_sum` / `mi355gp_predict_sum` and through the host classes, against golden vectors from the
reference's own Add / White / Bias code.  Tolerances as for single kernels."""
import numpy as np
import pytest

import gpy_amd
from gpy_amd import _lib as L
from oracle import gp_oracle as O
from test_oracle_sum import load_sum_golden, sum_golden_names

pytestmark = pytest.mark.gpu
TOL_LML, TOL_ALPHA, TOL_GRAD = 1e-10, 1e-9, 1e-8


def _specs(g):
    out = []
    for kind, ARD, var, ls, dims, term in g["parts"]:
        th = np.array([var]) if ls is None else np.concatenate([[var], np.atleast_1d(ls)])
        out.append((kind, ARD, th, dims, term))
    return out


@pytest.mark.parametrize("name", sum_golden_names())
def test_sum_golden_through_the_c_abi(name):
    g = load_sum_golden(name)
    c = L.Context(0)
    try:
        c.set_data(g["X"], g["Y"])
        info, r = c.exact_inference_sum(_specs(g), g["noise"], want_diag=True)
        assert info == 0
        assert abs(r["lml"] - g["lml"]) <= TOL_LML * abs(g["lml"])
        assert np.linalg.norm(r["alpha"] - g["alpha"]) <= TOL_ALPHA * np.linalg.norm(g["alpha"])
        assert np.abs(r["dtheta"] - g["dtheta"]).max() <= TOL_GRAD * np.abs(g["dtheta"]).max()
        assert abs(r["dnoise"] - g["dnoise"]) <= TOL_GRAD * abs(g["dnoise"])
        K = c.fetch(L.FETCH_K)
        assert np.abs(K[0] - g["K_row0"]).max() <= 1e-13
        mu, var = c.predict_sum(_specs(g), g["Xs"])
        assert np.abs(mu - g["pred_mu"]).max() <= 1e-9 and np.abs(var - g["pred_var"]).max() <= 1e-9
        _, cov = c.predict_sum(_specs(g), g["Xs"], full_cov=True)
        assert np.abs(cov - g["pred_cov"]).max() <= 1e-9
    finally:
        c.close()


def test_add_kernel_host_classes_in_gpregression():
    g = load_sum_golden("sum_n200_rbfard02_m52_white_bias")
    k = (gpy_amd.RBF(2, variance=1.3, lengthscale=[0.7, 1.1], ARD=True, active_dims=[0, 2])
         + gpy_amd.Matern52(3, variance=0.6, lengthscale=1.5) + gpy_amd.White(3, 0.05) + gpy_amd.Bias(3, 0.4))
    m = gpy_amd.GPRegression(g["X"], g["Y"], k, noise_var=g["noise"])
    assert abs(m.log_likelihood() - g["lml"]) <= TOL_LML * abs(g["lml"])
    gref = np.concatenate([g["dtheta"], [g["dnoise"]]])
    assert np.abs(m.gradient - gref).max() <= TOL_GRAD * np.abs(gref).max()
    mu, var = m.predict_noiseless(g["Xs"])
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-9 and np.abs(var - g["pred_var"]).max() <= 1e-9
    # generic path: a foreign dL_dK goes part by part through the C-ABI / host formulas (add.py:81-82)
    A = np.random.default_rng(0).standard_normal((200, 200))
    k.update_gradients_full(A, g["X"])
    parts = g["parts"]
    ref = []
    for kind, ARD, var_, ls, dims, _term in parts:
        if kind == "white":
            ref.append([np.trace(A)])
        elif kind == "bias":
            ref.append([A.sum()])
        else:
            dv, dl = O.update_gradients_full(kind, A, g["X"][:, dims], None, var_, ls, ARD)
            ref.append(np.concatenate([[dv], np.atleast_1d(dl)]))
    ref = np.concatenate(ref)
    assert np.abs(k.gradient - ref).max() <= TOL_GRAD * np.abs(ref).max()
    f0 = m.objective_function()
    m.optimize(max_iters=5)
    assert m.objective_function() < f0


def test_prod_kernel_host_classes_in_gpregression():
    """`Prod` / `Add` of `Prod` through the host classes: fused device path vs the reference's golden vectors, and the
    generic (foreign dL_dK) path of Prod.update_gradients_full (prod.py:377-385) vs the oracle."""
    g = load_sum_golden("prod_n230_three_factors_plus_rbf_dy2")
    k = (gpy_amd.RBF(1, variance=0.9, lengthscale=1.2, active_dims=[0])
         * gpy_amd.Exponential(1, variance=1.2, lengthscale=2.5, active_dims=[1]) * gpy_amd.Bias(3, 0.6)
         + gpy_amd.RBF(3, variance=0.5, lengthscale=[0.8, 1.0, 1.7], ARD=True))
    assert isinstance(k, gpy_amd.Add) and isinstance(k.parts[0], gpy_amd.Prod) and len(k.parts[0].parts) == 3
    m = gpy_amd.GPRegression(g["X"], g["Y"], k, noise_var=g["noise"])
    assert abs(m.log_likelihood() - g["lml"]) <= TOL_LML * abs(g["lml"])
    gref = np.concatenate([g["dtheta"], [g["dnoise"]]])
    assert np.abs(m.gradient - gref).max() <= TOL_GRAD * np.abs(gref).max()
    mu, var = m.predict_noiseless(g["Xs"])
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-9 and np.abs(var - g["pred_var"]).max() <= 1e-9
    n = g["X"].shape[0]
    A = np.random.default_rng(1).standard_normal((n, n))
    k.update_gradients_full(A, g["X"])
    ref, parts = [], g["parts"]
    for grp in O._terms(parts):
        for i in grp:
            W = A
            for j in grp:
                if j != i:
                    W = W * O._part_K(parts[j], g["X"])
            ref.append((i, O._part_grads(parts[i], W, g["X"])))
    ref = np.concatenate([r for _, r in sorted(ref, key=lambda t: t[0])])
    assert np.abs(k.gradient - ref).max() <= TOL_GRAD * np.abs(ref).max()
    assert np.abs(k.K(g["X"])[0] - g["K_row0"]).max() <= 1e-13
    f0 = m.objective_function()
    m.optimize(max_iters=5)
    assert m.objective_function() < f0


def test_covariance_between_points_and_full_cov_prediction_with_a_sum_of_products():
    """Posterior.covariance_between_points (posterior.py:109-130) and the full-covariance prediction for an `Add` of a
    `Prod` and plain parts, against the oracle's expression evaluation (cross-covariances: White contributes nothing)."""
    from scipy.linalg import solve_triangular
    g = load_sum_golden("prod_n260_rbfard_x_m52_plus_white")
    parts, X = g["parts"], g["X"]
    r = O.sum_parameters_changed(parts, X, g["Y"], g["noise"])
    rng = np.random.default_rng(11)
    X1, X2 = rng.standard_normal((23, X.shape[1])), rng.standard_normal((140, X.shape[1]))
    t1 = solve_triangular(r["L"], O.sum_kern_K(parts, X, X1), lower=True)
    t2 = solve_triangular(r["L"], O.sum_kern_K(parts, X, X2), lower=True)
    expect = O.sum_kern_K(parts, X1, X2) - t1.T @ t2
    c = L.Context(0)
    try:
        c.set_data(X, g["Y"])
        info, _ = c.exact_inference_sum(_specs(g), g["noise"])
        assert info == 0
        got = c.covariance_between_points(_specs(g), X1, X2)
        assert got.shape == (23, 140) and np.abs(got - expect).max() <= 1e-10
        mu, cov = c.predict_sum(_specs(g), X1, full_cov=True)
        cov_ref = O.sum_kern_K(parts, X1) - t1.T @ t1                       # K(X1, X1): White on its diagonal (static.py:77-81)
        assert np.abs(cov - cov_ref).max() <= 1e-10
    finally:
        c.close()


def test_studentt_process_inference_matches_reference_golden():
    """ExactStudentTInference on the device (C-ABI mi355gp_exact_studentt_sum) vs the reference's own outputs.
    There is no noise term: Ky = K + 1e-8 I has cond ~ 1e9..1e10, so two correct fp64 factorisations differ by
    ~cond*eps: LML rel 1e-6, alpha / dL_dm / gradients rel 1e-4 (the same inputs with noise pass at 1e-10, see the
    Gaussian tests; the CPU oracle matches the reference to 1e-11 only because both call the same LAPACK)."""
    from test_oracle_studentt import load_studentt_golden, studentt_golden_names
    for name in studentt_golden_names():
        g = load_studentt_golden(name)
        D = g["X"].shape[1]
        cls = {"rbf": gpy_amd.RBF, "matern52": gpy_amd.Matern52}[g["kind"]]
        k = cls(D, variance=g["variance"], lengthscale=g["ls"], ARD=g["ARD"])
        inf = gpy_amd.ExactStudentTInference()
        post, lml, gd = inf.inference(k, g["X"], g["Y"], g["nu"])
        print(name, abs(lml - g["lml"]) / abs(g["lml"]),
              np.linalg.norm(post.woodbury_vector - g["alpha"]) / np.linalg.norm(g["alpha"]))
        assert abs(lml - g["lml"]) <= 1e-6 * abs(g["lml"])
        assert np.linalg.norm(post.woodbury_vector - g["alpha"]) <= 1e-4 * np.linalg.norm(g["alpha"])
        assert abs(gd["dL_dnu"] - g["dL_dnu"]) <= 1e-5 * abs(g["dL_dnu"])
        assert np.linalg.norm(gd["dL_dm"] - g["dL_dm"]) <= 1e-4 * np.linalg.norm(g["dL_dm"])
        k.update_gradients_full(gd["dL_dK"], g["X"])
        assert np.abs(k.gradient - g["dtheta"]).max() <= 1e-4 * np.abs(g["dtheta"]).max()
        # materialised dL_dK carries the Student-t factor on its alpha alpha^T term (exact_studentt_inference.py:46)
        G = np.asarray(gd["dL_dK"])
        assert np.abs(G[g["rows"]] - g["dL_dK_rows"]).max() <= 1e-4 * np.abs(g["dL_dK_rows"]).max()
        k2 = cls(D, variance=g["variance"], lengthscale=g["ls"], ARD=g["ARD"])
        k2.update_gradients_full(G, g["X"])                       # the non-fused consumer sees the same gradients
        assert np.abs(k2.gradient - g["dtheta"]).max() <= 1e-4 * np.abs(g["dtheta"]).max()
        # StudentTPosterior._raw_predict: predictive (co)variance scaled by (nu+beta-2)/(nu+N-2) (posterior.py:338-349)
        assert isinstance(post, gpy_amd.StudentTPosterior)
        mu, var = post._raw_predict(k, g["Xs"], g["X"])
        _, cov = post._raw_predict(k, g["Xs"], g["X"], full_cov=True)
        assert np.abs(mu - g["pred_mu"]).max() <= 1e-4 * np.abs(g["pred_mu"]).max()
        assert np.abs(var - g["pred_var"]).max() <= 1e-4 * np.abs(g["pred_var"]).max()
        assert np.abs(cov - g["pred_cov"]).max() <= 1e-4 * np.abs(g["pred_cov"]).max()
        # self-consistency that does not suffer from the conditioning: Ky alpha = Y through the device kernel matrix
        Ky = k.K(g["X"]) + 1e-8 * np.eye(g["X"].shape[0])
        assert np.abs(Ky @ post.woodbury_vector - g["Y"]).max() <= 1e-6

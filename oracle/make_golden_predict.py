"""TEST INFRASTRUCTURE ONLY -- tests/golden/predict_*.npz FROM THE REFERENCE'S OWN CODE: the prediction-side callers of the
exact path in `GPy/core/gp.py`, evaluated with the reference's kernel / posterior / likelihood objects through
oracle/ref_loader.py (the `GP` class itself needs paramz's Model machinery, so its method bodies are restated line by line
around those objects):
  predict_quantiles        gp.py:395-416  -> Gaussian.predictive_quantiles (gaussian.py:118-119)
  log_predictive_density   gp.py:700-714  -> Gaussian.log_predictive_density (gaussian.py:329-334)
  predictive_gradients     gp.py:418-474  -> kern.gradients_X / gradients_X_diag (stationary.py:245-252,330-361)
  posterior_covariance_between_points gp.py:749-790 (with the likelihood term)

    python oracle/make_golden_predict.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import default_theta, synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def case(ns, name, kind, N, D, ARD, Dy=1, seed=0, M=37):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    var, ls, noise = default_theta(D, ARD)
    k = ref_loader.make_kernel(ns, kind, D, var, ls if ARD else float(np.atleast_1d(ls)[0]), ARD)
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    rng = np.random.default_rng(seed + 70)
    Xs = rng.standard_normal((M, D))
    ys = rng.standard_normal((M, Dy))
    # gp.py:290-306 _raw_predict
    mu, v = post._raw_predict(k, Xs, X, full_cov=False)
    # gp.py:408-416 predict_quantiles
    q = lik.predictive_quantiles(mu, v, (2.5, 50.0, 97.5))
    # gp.py:712-714 log_predictive_density
    lpd = lik.log_predictive_density(ys, mu, v)
    # gp.py:440-474 predictive_gradients (woodbury_inv.ndim == 2 branch; no normaliser)
    mean_jac = np.empty((M, D, Dy))
    for i in range(Dy):
        mean_jac[:, :, i] = k.gradients_X(post.woodbury_vector[:, i:i + 1].T, Xs, X)
    dv_dX = k.gradients_X_diag(np.ones(M), Xs)
    alpha = -2.0 * np.dot(k.K(Xs, X), post.woodbury_inv)
    var_jac = dv_dX + k.gradients_X(alpha, Xs, X)
    # gp.py:749-790 posterior_covariance_between_points with the likelihood (predictive_values, gaussian.py:102-110)
    X1, X2 = Xs[:9], Xs[:9]
    cov = post.covariance_between_points(k, X, X1, X2)
    m1, _ = post._raw_predict(k, X1, X, full_cov=True)
    _, cov_lik = lik.predictive_values(m1, cov.copy(), full_cov=True)     # predictive_values adds the noise IN PLACE
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind=kind, ARD=ARD, X=X, Y=Y, variance=var,
                        lengthscale=np.atleast_1d(ls), noise=noise, Xs=Xs, ys=ys, mu=np.asarray(mu), var=np.asarray(v),
                        quantiles=np.stack([np.asarray(a) for a in q]), lpd=np.asarray(lpd), mean_jac=mean_jac,
                        var_jac=np.asarray(var_jac), cov=np.asarray(cov), cov_lik=np.asarray(cov_lik), lml=float(lml))
    print("%-40s lml=% .12e |mean_jac|=%.4e |var_jac|=%.4e" % (name, lml, np.abs(mean_jac).max(), np.abs(var_jac).max()))


def sparse_case(ns, name, kind, N, Mi, D, ARD, Dy=1, seed=0, M=29):
    """The same callers for the sparse model: `_predictive_variable` is Z (core/sparse_gp.py:72-74), the posterior is VarDTC's."""
    import importlib
    from oracle.sparse_oracle import synthetic_Z
    vd = importlib.import_module("GPy.inference.latent_function_inference.var_dtc")
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    Z = synthetic_Z(X, Mi, seed)
    var, ls, noise = default_theta(D, ARD)
    k = ref_loader.make_kernel(ns, kind, D, var, ls if ARD else float(np.atleast_1d(ls)[0]), ARD)
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = vd.VarDTC().inference(k, X, Z, lik, Y)
    rng = np.random.default_rng(seed + 80)
    Xs = rng.standard_normal((M, D))
    ys = rng.standard_normal((M, Dy))
    mu, v = post._raw_predict(k, Xs, Z, full_cov=False)
    q = lik.predictive_quantiles(mu, v, (2.5, 97.5))
    lpd = lik.log_predictive_density(ys, mu, v)
    mean_jac = np.empty((M, D, Dy))
    for i in range(Dy):
        mean_jac[:, :, i] = k.gradients_X(post.woodbury_vector[:, i:i + 1].T, Xs, Z)
    alpha = -2.0 * np.dot(k.K(Xs, Z), post.woodbury_inv)
    var_jac = k.gradients_X_diag(np.ones(M), Xs) + k.gradients_X(alpha, Xs, Z)
    # (Posterior.covariance_between_points refuses woodbury_inv-built posteriors, posterior.py:118-120: no golden for it)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind=kind, ARD=ARD, X=X, Y=Y, Z=Z, variance=var,
                        lengthscale=np.atleast_1d(ls), noise=noise, Xs=Xs, ys=ys, mu=np.asarray(mu), var=np.asarray(v),
                        quantiles=np.stack([np.asarray(a) for a in q]), lpd=np.asarray(lpd), mean_jac=mean_jac,
                        var_jac=np.asarray(var_jac), lml=float(np.asarray(lml).ravel()[0]))
    print("%-40s lml=% .12e |mean_jac|=%.4e |var_jac|=%.4e" % (name, float(np.asarray(lml).ravel()[0]), np.abs(mean_jac).max(),
                                                            np.abs(var_jac).max()))


def main():
    ns = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    sparse_case(ns, "predsparse_n600_m40_d3_rbf_ard", "rbf", 600, 40, 3, True, seed=4)
    sparse_case(ns, "predsparse_n500_m33_d2_matern52_iso_dy2", "matern52", 500, 33, 2, False, Dy=2, seed=5)
    case(ns, "predict_n400_d3_rbf_iso", "rbf", 400, 3, False)
    case(ns, "predict_n333_d5_matern52_ard_dy2", "matern52", 333, 5, True, Dy=2, seed=2)
    case(ns, "predict_n280_d2_matern32_ard", "matern32", 280, 2, True, seed=3)


if __name__ == "__main__":
    main()

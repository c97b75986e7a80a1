"""TEST INFRASTRUCTURE ONLY -- tests/golden/studentt_*.npz FROM THE REFERENCE'S OWN CODE:
`ExactStudentTInference.inference` (GPy/inference/latent_function_inference/exact_studentt_inference.py) +
`Stationary.update_gradients_full`, executed through oracle/ref_loader.py.

    python oracle/make_golden_studentt.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import default_theta, synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def case(ns, mod, name, kind, N, D, ARD, nu, Dy=1, seed=0):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    var, ls, _ = default_theta(D, ARD)
    k = ref_loader.make_kernel(ns, kind, D, var, ls if ARD else float(np.atleast_1d(ls)[0]), ARD)
    post, lml, gd = mod.ExactStudentTInference().inference(k, X, Y, nu)
    k.update_gradients_full(gd["dL_dK"], X)
    # StudentTPosterior._raw_predict (posterior.py:338-349): Gaussian predictive variance scaled by (nu+beta-2)/(nu+N-2)
    Xs = np.random.default_rng(seed + 50).standard_normal((23, D))
    pred_mu, pred_var = post._raw_predict(k, Xs, X, full_cov=False)
    _, pred_cov = post._raw_predict(k, Xs, X, full_cov=True)
    rows = np.sort(np.random.default_rng(seed + 51).choice(N, 8, replace=False))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind=kind, ARD=ARD, X=X, Y=Y, variance=var,
                        lengthscale=np.atleast_1d(ls), nu=nu, lml=float(lml), alpha=np.asarray(post.woodbury_vector),
                        dL_dnu=float(gd["dL_dnu"]), dL_dm=np.asarray(gd["dL_dm"]), Xs=Xs, pred_mu=np.asarray(pred_mu),
                        pred_var=np.asarray(pred_var), pred_cov=np.asarray(pred_cov), rows=rows,
                        dL_dK_rows=np.asarray(gd["dL_dK"])[rows],
                        dtheta=np.concatenate([np.atleast_1d(np.asarray(k.variance.gradient, float)),
                                               np.atleast_1d(np.asarray(k.lengthscale.gradient, float))]))
    print("%-34s lml=% .12e" % (name, lml))


def main():
    ns = ref_loader.load()
    sys.modules["GPy.inference.latent_function_inference"].LatentFunctionInference  # the stub base class exists
    post_mod = importlib.import_module("GPy.inference.latent_function_inference.posterior")
    assert hasattr(post_mod, "StudentTPosterior")
    mod = importlib.import_module("GPy.inference.latent_function_inference.exact_studentt_inference")
    os.makedirs(OUT, exist_ok=True)
    case(ns, mod, "studentt_n300_d3_rbf_iso_nu5", "rbf", 300, 3, False, 5.0)
    case(ns, mod, "studentt_n257_d4_matern52_ard_nu3p5_dy2", "matern52", 257, 4, True, 3.5, Dy=2, seed=1)


if __name__ == "__main__":
    main()

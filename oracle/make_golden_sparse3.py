"""TEST INFRASTRUCTURE ONLY -- tests/golden/sparse3_*.npz FROM THE REFERENCE'S OWN CODE (`VarDTC.inference` +
`SparseGP._update_gradients`, run in place through oracle/ref_loader.py) for the cases round 2 left out:
per-point noise variances with SEVERAL output columns (var_dtc.py:84-129,240-256) and more than 32 input dimensions.

    python oracle/make_golden_sparse3.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import synthetic  # noqa: E402
from oracle.make_golden_prod import assemble  # noqa: E402
from oracle.make_golden_sparse2 import OUT, _Lik, case  # noqa: E402
from oracle.sparse_oracle import synthetic_Z  # noqa: E402


def case_prod(ns, name, specs, N, M, D, noise, Dy=1, seed=0):
    """specs: [(kind, ARD, variance, lengthscale, active_dims, term)]; parts sharing a non-zero term are the factors of one
    `Prod` (GPy/kern/src/prod.py).  Same evaluation and file layout as make_golden_sparse2.case, plus `terms`."""
    vd = importlib.import_module("GPy.inference.latent_function_inference.var_dtc")
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    Z = synthetic_Z(X, M, seed)
    k, leaves = assemble(ns, specs, D)

    def grads():
        out = []
        for p in leaves:
            out.append(np.atleast_1d(np.asarray(p.variance.gradient, float)))
            if hasattr(p, "lengthscale"):
                out.append(np.atleast_1d(np.asarray(p.lengthscale.gradient, float)))
        return np.concatenate(out)
    post, lml, gd = vd.VarDTC().inference(k, X, Z, _Lik(noise), Y)
    k.update_gradients_diag(gd["dL_dKdiag"], X)                       # sparse_gp.py:110-115
    g = grads().copy()
    k.update_gradients_full(gd["dL_dKnm"], X, Z)
    g += grads()
    k.update_gradients_full(gd["dL_dKmm"], Z, None)
    g += grads()
    dZ = k.gradients_X(gd["dL_dKmm"], Z) + k.gradients_X(gd["dL_dKnm"].T, Z, X)      # :116-118
    Xs = np.random.default_rng(seed + 9).standard_normal((31, D))
    mu, var = post._raw_predict(k, Xs, Z, full_cov=False)
    _, cov = post._raw_predict(k, Xs, Z, full_cov=True)
    rows = np.sort(np.random.default_rng(seed + 3).choice(N, 5, replace=False))
    flat = dict(kinds=np.array([s[0] for s in specs]), ARDs=np.array([bool(s[1]) for s in specs]),
                variances=np.array([float(s[2]) for s in specs]), terms=np.array([int(s[5]) for s in specs]))
    for i, s in enumerate(specs):
        flat["ls%d" % i] = np.atleast_1d(np.asarray(s[3], float)) if s[3] is not None else np.zeros(0)
        flat["dims%d" % i] = np.asarray(s[4] if s[4] is not None else list(range(D)), int)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, Y=Y, Z=Z, noise=np.atleast_1d(np.asarray(noise, float)),
                        mean_w=np.zeros((0, 0)), lml=float(np.asarray(lml).ravel()[0]), dtheta=g,
                        dnoise=np.asarray(gd["dL_dthetaL"], float).ravel(), dZ=np.asarray(dZ),
                        woodbury_vector=np.asarray(post.woodbury_vector), woodbury_inv=np.asarray(post.woodbury_inv),
                        dL_dm=np.asarray(gd["dL_dm"]), rows=rows, dL_dKnm_rows=np.asarray(gd["dL_dKnm"])[rows],
                        dL_dKmm=np.asarray(gd["dL_dKmm"]), Xs=Xs, pred_mu=np.asarray(mu), pred_var=np.asarray(var),
                        pred_cov=np.asarray(cov), **flat)
    print("%-44s lml=% .12e |dZ|=%.6e" % (name, float(np.asarray(lml).ravel()[0]), np.linalg.norm(dZ)))


def main():
    ns = ref_loader.load()
    ref_loader.load_sum_kernels(ns)
    rng = np.random.default_rng(5)
    case(ns, "sparse3_rbf_hetero_dy2_n380_m28_d3", [("rbf", True, 1.2, [0.8, 1.1, 1.5], [0, 1, 2])], 380, 28, 3,
         0.04 + 0.1 * rng.random(380), Dy=2, seed=6)
    case(ns, "sparse3_m52_bias_hetero_dy3_n300_m35_d2", [("matern52", False, 0.9, 1.2, [0, 1]), ("bias", False, 0.2, None, [0, 1])],
         300, 35, 2, 0.05 + 0.08 * rng.random(300), Dy=3, seed=7)
    D = 40
    case(ns, "sparse3_rbf_ard_d40_n700_m48", [("rbf", True, 1.3, list(np.linspace(0.5, 2.0, D) * np.sqrt(D / 8.0)), list(range(D)))],
         700, 48, D, 0.1, seed=8)
    # product kernels in the sparse path (prod.py:58-113)
    case_prod(ns, "sparse3_prod_rbf01_x_m32_2_n420_m30_d3",
              [("rbf", False, 1.3, [0.9], [0, 1], 1), ("matern32", False, 0.8, [1.4], [2], 1)], 420, 30, 3, 0.1, seed=9)
    case_prod(ns, "sparse3_prod_rbfard_x_m52_plus_white_plus_rbf_n460_m36_d4_dy2",
              [("rbf", True, 1.1, [0.7, 1.3], [0, 2], 1), ("matern52", False, 0.7, [1.6], [1, 3], 1),
               ("white", False, 0.05, None, [0, 1, 2, 3], 0), ("rbf", False, 0.5, [1.1], [0, 1, 2, 3], 0)], 460, 36, 4, 0.15, Dy=2, seed=10)
    # everything round 3 added at once: a product of two ARD factors on 20 + 20 of 40 input dimensions plus a White term,
    # per-point noise, two output columns
    rng2 = np.random.default_rng(12)
    D2 = 40
    case_prod(ns, "sparse3_combo_prod_ard_d40_white_hetero_dy2_n520_m44",
              [("rbf", True, 1.2, list(np.linspace(0.9, 2.2, 20) * np.sqrt(20 / 6.0)), list(range(0, 20)), 1),
               ("matern32", True, 0.8, list(np.linspace(1.1, 2.6, 20) * np.sqrt(20 / 6.0)), list(range(20, 40)), 1),
               ("white", False, 0.03, None, list(range(D2)), 0)], 520, 44, D2, 0.05 + 0.1 * rng2.random(520), Dy=2, seed=12)
    case_prod(ns, "sparse3_prod_three_factors_bias_hetero_n350_m25_d3",
              [("rbf", False, 0.9, [1.2], [0], 2), ("exponential", False, 1.2, [2.5], [1], 2), ("bias", False, 0.6, None, [0, 1, 2], 2),
               ("matern52", True, 0.5, [0.8, 1.0, 1.7], [0, 1, 2], 0)], 350, 25, 3, 0.05 + 0.1 * rng.random(350), seed=11)


if __name__ == "__main__":
    main()

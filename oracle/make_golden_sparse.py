"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/sparse_*.npz FROM THE REFERENCE'S OWN CODE:
the unmodified `VarDTC.inference` (GPy/inference/latent_function_inference/var_dtc.py) followed by the kernel /
inducing-input gradient assembly of `SparseGP._update_gradients` (GPy/core/sparse_gp.py:108-118), executed through
oracle/ref_loader.py on seeded synthetic inputs.

    python oracle/make_golden_sparse.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import default_theta, synthetic  # noqa: E402
from oracle.sparse_oracle import synthetic_Z  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_reference(ns, kind, X, Z, Y, var, ls, ARD, noise):
    vd = importlib.import_module("GPy.inference.latent_function_inference.var_dtc")
    k = ref_loader.make_kernel(ns, kind, X.shape[1], var, ls if ARD else float(np.atleast_1d(ls)[0]), ARD)
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = vd.VarDTC().inference(k, X, Z, lik, Y)

    def grads():
        return np.concatenate([np.atleast_1d(np.asarray(k.variance.gradient, float)),
                               np.atleast_1d(np.asarray(k.lengthscale.gradient, float))])
    k.update_gradients_diag(gd["dL_dKdiag"], X)
    g = grads().copy()
    k.update_gradients_full(gd["dL_dKnm"], X, Z)
    g += grads()
    k.update_gradients_full(gd["dL_dKmm"], Z, None)
    g += grads()
    dZ = k.gradients_X(gd["dL_dKmm"], Z) + k.gradients_X(gd["dL_dKnm"].T, Z, X)
    return dict(lml=float(np.asarray(lml).ravel()[0]), dtheta=g, dnoise=float(np.asarray(gd["dL_dthetaL"]).ravel()[0]),
                dZ=np.asarray(dZ), woodbury_vector=np.asarray(post.woodbury_vector),
                woodbury_inv=np.asarray(post.woodbury_inv), dL_dKmm=np.asarray(gd["dL_dKmm"]))


def case(ns, name, kind, N, M, D, ARD, Dy=1, seed=0):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    Z = synthetic_Z(X, M, seed)
    var, ls, noise = default_theta(D, ARD)
    r = run_reference(ns, kind, X, Z, Y, var, ls, ARD, noise)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind=kind, ARD=ARD, X=X, Y=Y, Z=Z, variance=var,
                        lengthscale=np.atleast_1d(ls), noise=noise, **r)
    print("%-36s lml=% .12e |dZ|=%.6e" % (name, r["lml"], np.linalg.norm(r["dZ"])))


def main():
    ns = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    case(ns, "sparse_n300_m20_d3_rbf_iso", "rbf", 300, 20, 3, False)
    case(ns, "sparse_n500_m48_d4_rbf_ard", "rbf", 500, 48, 4, True, seed=1)
    case(ns, "sparse_n400_m33_d2_matern52_ard_dy2", "matern52", 400, 33, 2, True, Dy=2, seed=2)
    case(ns, "sparse_n350_m25_d3_matern32_iso", "matern32", 350, 25, 3, False, seed=3)
    case(ns, "sparse_n260_m17_d2_exponential_iso", "exponential", 260, 17, 2, False, seed=4)
    case(ns, "sparse_n1500_m130_d6_rbf_iso", "rbf", 1500, 130, 6, False, seed=5)


if __name__ == "__main__":
    main()

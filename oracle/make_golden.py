"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz FROM THE REFERENCE'S OWN CODE.

Runs the reference's unmodified `ExactGaussianInference.inference` + `Gaussian.update_gradients` +
`Stationary.update_gradients_full` (the `GP.parameters_changed` sequence, reference
`GPy/core/gp.py:278-280`) through `oracle/ref_loader.py` on seeded synthetic inputs and stores
inputs and outputs.  The fixtures travel to the GPU box, where /root/reference does not exist.

    python oracle/make_golden.py            # rewrites tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import synthetic, default_theta  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def case(ns, name, kind, N, D, ARD, Dy=1, seed=0, noise_vec=False, theta=None, predict=False):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    var, ls, noise = default_theta(D, ARD) if theta is None else theta
    k = ref_loader.make_kernel(ns, kind, D, var, ls if ARD else float(np.atleast_1d(ls)[0]), ARD)
    lik = ns.Gaussian(variance=noise)
    rng = np.random.default_rng(seed + 100)
    if noise_vec:
        nv = noise * (0.5 + rng.random(N))
        post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y, variance=nv)
    else:
        nv = np.array([noise])
        post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    K = np.asarray(post._K)
    L = np.asarray(post.woodbury_chol)
    G = np.asarray(gd["dL_dK"])
    rows = np.array(sorted({0, N // 3, N // 2, N - 1}))
    d = dict(kind=kind, ARD=ARD, X=X, Y=Y, variance=var, lengthscale=np.atleast_1d(ls), noise=nv,
             lml=float(lml), alpha=np.asarray(post.woodbury_vector),
             dvar=np.asarray(k.variance.gradient, float), dlen=np.asarray(k.lengthscale.gradient, float),
             dnoise=np.asarray(lik.variance.gradient, float), diag_dL_dK=np.diag(G).copy(),
             logdet=2.0 * np.sum(np.log(np.diag(L))), rows=rows, K_rows=K[rows], L_rows=L[rows],
             dL_dK_rows=G[rows], Kdiag=np.asarray(k.Kdiag(X)))
    if N <= 64:
        d.update(K=K, L=L, dL_dK=G)
    # generic update_gradients_full with a caller-supplied, non-symmetric dL_dK and rectangular X2
    # (X2, A are regenerated in the tests from seedA: keeps the fixtures small)
    M = max(3, N // 3)
    seedA = seed + 1000
    rngA = np.random.default_rng(seedA)
    X2 = rngA.standard_normal((M, D))
    A = rngA.standard_normal((N, M))
    k.update_gradients_full(A, X, X2)
    d.update(seedA=seedA, M=M, K_X_X2_rows=np.asarray(k.K(X, X2))[rows],
             dvar_A=np.asarray(k.variance.gradient, float), dlen_A=np.asarray(k.lengthscale.gradient, float))
    if predict:
        Xs = rng.standard_normal((17, D))
        mu, v = post._raw_predict(k, Xs, pred_var=X, full_cov=False)
        mu2, V = post._raw_predict(k, Xs, pred_var=X, full_cov=True)
        d.update(Xs=Xs, pred_mu=mu, pred_var=v, pred_cov=V)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("%-34s lml=% .12e" % (name, lml))


def main():
    ns = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    for kind in ("rbf", "matern52", "matern32", "exponential"):
        for ARD in (False, True):
            tag = "%s_%s" % (kind, "ard" if ARD else "iso")
            case(ns, "n64_d3_" + tag, kind, 64, 3, ARD, predict=True)
            case(ns, "n512_d2_" + tag, kind, 512, 2, ARD)
    case(ns, "n64_d3_rbf_iso_dy3", "rbf", 64, 3, False, Dy=3)
    case(ns, "n200_d5_matern52_ard_dy3", "matern52", 200, 5, True, Dy=3, seed=1)
    case(ns, "n64_d3_rbf_iso_hetero", "rbf", 64, 3, False, noise_vec=True)
    case(ns, "n333_d4_matern32_ard_hetero", "matern32", 333, 4, True, noise_vec=True, seed=2)
    case(ns, "n1_d2_rbf_iso", "rbf", 1, 2, False)
    case(ns, "n7_d1_exponential_iso", "exponential", 7, 1, False)
    case(ns, "n1000_d8_rbf_iso", "rbf", 1000, 8, False, seed=3)
    case(ns, "n1000_d8_matern52_ard", "matern52", 1000, 8, True, seed=4)


if __name__ == "__main__":
    main()

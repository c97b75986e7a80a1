"""TEST INFRASTRUCTURE ONLY -- tests/golden/sum_*.npz FROM THE REFERENCE'S OWN CODE: `GPy.kern.Add` of stationary,
White and Bias parts (GPy/kern/src/add.py, static.py) through `ExactGaussianInference.inference`,
`Add.update_gradients_full` and `PosteriorExact._raw_predict`, executed by oracle/ref_loader.py.

    python oracle/make_golden_sum.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def build(ns, spec, D):
    kind, ARD, var, ls, dims = spec
    if kind == "white":
        return ns.White(D, variance=var)
    if kind == "bias":
        return ns.Bias(D, variance=var)
    cls = getattr(ns, ref_loader.KERNELS[kind])
    return cls(len(dims), variance=var, lengthscale=ls if ARD else float(np.atleast_1d(ls)[0]), ARD=ARD,
               active_dims=list(dims))


def case(ns, name, N, D, specs, noise=0.1, Dy=1, seed=0):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    k = ns.Add([build(ns, s, D) for s in specs])
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    g = []
    for p in k.parts:
        g.append(np.atleast_1d(np.asarray(p.variance.gradient, float)))
        if hasattr(p, "lengthscale"):
            g.append(np.atleast_1d(np.asarray(p.lengthscale.gradient, float)))
    Xs = np.random.default_rng(seed + 5).standard_normal((11, D))
    mu, var = post._raw_predict(k, Xs, pred_var=X, full_cov=False)
    _, cov = post._raw_predict(k, Xs, pred_var=X, full_cov=True)
    spec_json = json.dumps([[s[0], bool(s[1]), float(s[2]), [float(v) for v in np.atleast_1d(s[3])] if s[3] is not None
                             else None, [int(d) for d in s[4]] if s[4] is not None else None] for s in specs])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, Y=Y, noise=noise, specs=spec_json, lml=float(lml),
                        alpha=np.asarray(post.woodbury_vector), dtheta=np.concatenate(g),
                        dnoise=float(np.asarray(lik.variance.gradient).ravel()[0]), Xs=Xs, pred_mu=mu, pred_var=var,
                        pred_cov=cov, K_row0=np.asarray(k.K(X))[0])
    print("%-34s lml=% .12e" % (name, lml))


def main():
    ns = ref_loader.load_sum_kernels(ref_loader.load())
    os.makedirs(OUT, exist_ok=True)
    case(ns, "sum_n200_rbfard02_m52_white_bias", 200, 3,
         [("rbf", True, 1.3, [0.7, 1.1], [0, 2]), ("matern52", False, 0.6, [1.5], [0, 1, 2]),
          ("white", False, 0.05, None, None), ("bias", False, 0.4, None, None)])
    case(ns, "sum_n300_rbf_plus_white", 300, 2, [("rbf", False, 1.0, [0.9], [0, 1]), ("white", False, 0.2, None, None)],
         seed=1)
    case(ns, "sum_n257_m32ard_exp_bias_dy2", 257, 4,
         [("matern32", True, 0.9, [0.8, 1.2, 2.0, 0.6], [0, 1, 2, 3]), ("exponential", False, 0.3, [2.0], [1, 3]),
          ("bias", False, 1.1, None, None)], Dy=2, seed=2)


if __name__ == "__main__":
    main()

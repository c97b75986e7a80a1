"""TEST INFRASTRUCTURE ONLY -- tests/golden/prod_*.npz FROM THE REFERENCE'S OWN CODE: `GPy.kern.Prod` (and `Add` of
`Prod`s) of stationary / Bias / White factors (GPy/kern/src/prod.py, add.py, static.py) through
`ExactGaussianInference.inference`, `update_gradients_full` and `PosteriorExact._raw_predict`, executed by
oracle/ref_loader.py.  specs entries are [kind, ARD, variance, lengthscale, active_dims, term]: parts sharing a
non-zero term id are the factors of one `Prod`.

    python oracle/make_golden_prod.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import synthetic  # noqa: E402
from oracle.make_golden_sum import build  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def assemble(ns, specs, D):
    """the reference kernel object of a spec list + its leaf kernels in spec order"""
    leaves = [build(ns, s[:5], D) for s in specs]
    summands, seen = [], {}
    for s, k in zip(specs, leaves):
        t = s[5]
        if t == 0:
            summands.append([k])
        elif t in seen:
            seen[t].append(k)
        else:
            seen[t] = [k]
            summands.append(seen[t])
    tops = [g[0] if len(g) == 1 else ns.Prod(g) for g in summands]
    top = tops[0] if len(tops) == 1 else ns.Add(tops)

    def walk(k):                                   # Prod/Add copy their parts: collect the linked copies, in order
        return [q for p in k.parts for q in walk(p)] if hasattr(k, "parts") else [k]
    return top, walk(top)


def case(ns, name, N, D, specs, noise=0.1, Dy=1, seed=0):
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    k, leaves = assemble(ns, specs, D)
    assert len(leaves) == len(specs)
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    g = []
    for p in leaves:
        g.append(np.atleast_1d(np.asarray(p.variance.gradient, float)))
        if hasattr(p, "lengthscale"):
            g.append(np.atleast_1d(np.asarray(p.lengthscale.gradient, float)))
    Xs = np.random.default_rng(seed + 5).standard_normal((11, D))
    mu, var = post._raw_predict(k, Xs, pred_var=X, full_cov=False)
    _, cov = post._raw_predict(k, Xs, pred_var=X, full_cov=True)
    spec_json = json.dumps([[s[0], bool(s[1]), float(s[2]), [float(v) for v in np.atleast_1d(s[3])] if s[3] is not None
                             else None, [int(d) for d in s[4]] if s[4] is not None else None, int(s[5])] for s in specs])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, Y=Y, noise=noise, specs=spec_json, lml=float(lml),
                        alpha=np.asarray(post.woodbury_vector), dtheta=np.concatenate(g),
                        dnoise=float(np.asarray(lik.variance.gradient).ravel()[0]), Xs=Xs, pred_mu=mu, pred_var=var,
                        pred_cov=cov, K_row0=np.asarray(k.K(X))[0])
    print("%-40s lml=% .12e" % (name, lml))


def main():
    ns = ref_loader.load_sum_kernels(ref_loader.load())
    os.makedirs(OUT, exist_ok=True)
    case(ns, "prod_n200_rbf01_x_m32_2", 200, 3,
         [("rbf", False, 1.3, [0.9], [0, 1], 1), ("matern32", False, 0.8, [1.4], [2], 1)])
    case(ns, "prod_n260_rbfard_x_m52_plus_white", 260, 4,
         [("rbf", True, 1.1, [0.7, 1.3], [0, 2], 1), ("matern52", False, 0.7, [1.6], [1, 3], 1),
          ("white", False, 0.05, None, None, 0)], seed=1)
    case(ns, "prod_n230_three_factors_plus_rbf_dy2", 230, 3,
         [("rbf", False, 0.9, [1.2], [0], 2), ("exponential", False, 1.2, [2.5], [1], 2), ("bias", False, 0.6, None, None, 2),
          ("rbf", True, 0.5, [0.8, 1.0, 1.7], [0, 1, 2], 0)], Dy=2, seed=2)


if __name__ == "__main__":
    main()

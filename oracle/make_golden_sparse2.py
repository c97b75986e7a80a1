"""TEST INFRASTRUCTURE ONLY -- tests/golden/sparse2_*.npz FROM THE REFERENCE'S OWN CODE: `VarDTC.inference`
(var_dtc.py) + `SparseGP._update_gradients` (core/sparse_gp.py:108-118) + `Posterior._raw_predict` (posterior.py:198-262)
for the cases the round-1 fixtures did not cover: sum kernels (`Add` of stationary + White + Bias parts with
active_dims, add.py), per-point noise variances (heteroscedastic precision) and a mean function.

    python oracle/make_golden_sparse2.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gp_oracle import synthetic  # noqa: E402
from oracle.sparse_oracle import synthetic_Z  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class _Lik(object):
    """likelihood stand-in handing VarDTC a fixed (possibly per-point) Gaussian variance (what Gaussian /
    HeteroscedasticGaussian.gaussian_variance return, likelihoods/gaussian.py:69-70,361-362)"""

    def __init__(self, v):
        self.v = np.atleast_1d(np.asarray(v, float))
        self.size = self.v.size

    def gaussian_variance(self, Y_metadata=None):
        return self.v if self.v.size > 1 else self.v[0]

    def exact_inference_gradients(self, d, Y_metadata=None):
        return d


class _Mean(object):
    def __init__(self, w):
        self.w = w

    def f(self, X):
        return X @ self.w


def build_kernel(ns, spec, D):
    """spec: [(kind, ARD, variance, lengthscale, active_dims)] -> reference kernel (a single part or an Add)"""
    ks = []
    for kind, ARD, var, ls, dims in spec:
        if kind == "white":
            ks.append(ns.White(len(dims), variance=var, active_dims=list(dims)))
        elif kind == "bias":
            ks.append(ns.Bias(len(dims), variance=var, active_dims=list(dims)))
        else:
            cls = getattr(ns, ref_loader.KERNELS[kind])
            ks.append(cls(len(dims), variance=var, lengthscale=ls if ARD else float(np.atleast_1d(ls)[0]), ARD=ARD,
                          active_dims=list(dims)))
    return ks[0] if len(ks) == 1 else ns.Add(ks)


def leaves(k):
    return list(k.parts) if hasattr(k, "parts") else [k]


def grads(k):
    out = []
    for p in leaves(k):
        out.append(np.atleast_1d(np.asarray(p.variance.gradient, float)))
        if hasattr(p, "lengthscale"):
            out.append(np.atleast_1d(np.asarray(p.lengthscale.gradient, float)))
    return np.concatenate(out)


def case(ns, name, spec, N, M, D, noise, Dy=1, seed=0, mean_w=None):
    vd = importlib.import_module("GPy.inference.latent_function_inference.var_dtc")
    post_mod = importlib.import_module("GPy.inference.latent_function_inference.posterior")
    X, Y = synthetic(N, D, seed=seed, Dy=Dy)
    Z = synthetic_Z(X, M, seed)
    k = build_kernel(ns, spec, D)
    mf = None if mean_w is None else _Mean(mean_w)
    post, lml, gd = vd.VarDTC().inference(k, X, Z, _Lik(noise), Y, mean_function=mf)
    k.update_gradients_diag(gd["dL_dKdiag"], X)
    g = grads(k).copy()
    k.update_gradients_full(gd["dL_dKnm"], X, Z)
    g += grads(k)
    k.update_gradients_full(gd["dL_dKmm"], Z, None)
    g += grads(k)
    dZ = k.gradients_X(gd["dL_dKmm"], Z) + k.gradients_X(gd["dL_dKnm"].T, Z, X)
    Xs = np.random.default_rng(seed + 9).standard_normal((31, D))
    mu, var = post._raw_predict(k, Xs, Z, full_cov=False)
    _, cov = post._raw_predict(k, Xs, Z, full_cov=True)
    rows = np.sort(np.random.default_rng(seed + 3).choice(N, 5, replace=False))
    flat = dict(kinds=np.array([s[0] for s in spec]), ARDs=np.array([bool(s[1]) for s in spec]),
                variances=np.array([float(s[2]) for s in spec]))
    for i, s in enumerate(spec):
        flat["ls%d" % i] = np.atleast_1d(np.asarray(s[3], float)) if s[3] is not None else np.zeros(0)
        flat["dims%d" % i] = np.asarray(s[4], int)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, Y=Y, Z=Z, noise=np.atleast_1d(np.asarray(noise, float)),
                        mean_w=np.zeros((0, 0)) if mean_w is None else mean_w, lml=float(np.asarray(lml).ravel()[0]),
                        dtheta=g, dnoise=np.asarray(gd["dL_dthetaL"], float).ravel(), dZ=np.asarray(dZ),
                        woodbury_vector=np.asarray(post.woodbury_vector), woodbury_inv=np.asarray(post.woodbury_inv),
                        dL_dm=np.asarray(gd["dL_dm"]), rows=rows, dL_dKnm_rows=np.asarray(gd["dL_dKnm"])[rows],
                        dL_dKmm=np.asarray(gd["dL_dKmm"]), Xs=Xs, pred_mu=np.asarray(mu), pred_var=np.asarray(var),
                        pred_cov=np.asarray(cov), **flat)
    print("%-44s lml=% .12e |dZ|=%.6e" % (name, float(np.asarray(lml).ravel()[0]), np.linalg.norm(dZ)))


def main():
    ns = ref_loader.load()
    ref_loader.load_sum_kernels(ns)
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0)
    case(ns, "sparse2_rbf_white_n400_m30_d3", [("rbf", False, 1.3, 0.9, [0, 1, 2]), ("white", False, 0.05, None, [0, 1, 2])],
         400, 30, 3, 0.1)
    case(ns, "sparse2_m32ard_bias_rbf_dims_n500_m41_d4_dy2",
         [("matern32", True, 0.9, [0.8, 1.5], [0, 2]), ("bias", False, 0.3, None, [0, 1, 2, 3]),
          ("rbf", False, 0.6, 1.2, [1, 3])], 500, 41, 4, 0.2, Dy=2, seed=1)
    case(ns, "sparse2_rbf_hetero_n450_m33_d3", [("rbf", True, 1.1, [0.7, 1.0, 1.6], [0, 1, 2])], 450, 33, 3,
         0.05 + 0.1 * rng.random(450), seed=2)
    case(ns, "sparse2_m52_white_hetero_n300_m140_d2", [("matern52", False, 1.0, 1.1, [0, 1]), ("white", False, 0.02, None, [0, 1])],
         300, 140, 2, 0.03 + 0.05 * rng.random(300), seed=3)
    case(ns, "sparse2_rbf_meanfn_n350_m25_d3", [("rbf", False, 1.2, 1.0, [0, 1, 2])], 350, 25, 3, 0.15, seed=4,
         mean_w=np.array([[0.3], [-0.2], [0.1]]))


if __name__ == "__main__":
    main()

"""CPU: executes the reference-side binding `integration/gpy_mi355x.py` (the stub of INTEGRATION.md section 2) against the
REFERENCE'S OWN unmodified classes (oracle/ref_loader.py), with libmi355gp.so replaced by a mock whose entry points are
implemented by the CPU oracle and follow include/mi355gp.h's signatures.  Proves the reference-side glue -- argument
order, gradient installation into GPy's Param objects, lazy N x N fetches, the jitter ladder, the `K=` / foreign-kernel
route -- independently of the device code.  Skipped where /root/reference is absent (the GPU box)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import gp_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
sys.path.insert(0, os.path.join(ROOT, "integration"))
KINDS = ["rbf", "matern52", "matern32", "exponential"]


def _arr(ptr, n):
    return None if not ptr else np.ctypeslib.as_array(ptr, shape=(n,))


class OracleMockLib(object):
    """include/mi355gp.h entry points, computed by oracle/gp_oracle.py.  `fail_first`: number of leading inference calls
    that report `info = 3` (not positive definite) to drive the jitter ladder."""

    def __init__(self, fail_first=0):
        self.ctxs, self.fail_first, self.calls, self.extras = {}, fail_first, [], []

    def mi355gp_last_error(self):
        return b"mock"

    def mi355gp_create(self, device, ref):
        h = len(self.ctxs) + 1
        self.ctxs[h] = {}
        ref._obj.value = h
        return 0

    def mi355gp_set_data(self, ctx, X, N, D, R, Dy):
        assert X.shape == (N, D) and R.shape == (N, Dy) and X.flags.c_contiguous and R.flags.c_contiguous
        self.ctxs[ctx.value].update(X=X.copy(), R=R.copy())
        self.calls.append("set_data")
        return 0

    def _finish(self, c, res, out, alpha, diag):
        N, Dy = c["R"].shape
        out[:] = 0.0
        out[0], out[1], out[3] = res["lml"], res["logdet"], res["dL_dnoise"]
        if alpha:
            _arr(alpha, N * Dy)[:] = res["alpha"].ravel()
        if diag:
            _arr(diag, N)[:] = res["diag_dL_dK"]
        c["last"] = res

    def mi355gp_exact_inference(self, ctx, kind, ard, theta, noise, noise_len, jitter, extra, out, alpha, dtheta, diag, ms):
        c = self.ctxs[ctx.value]
        self.calls.append("exact_inference")
        self.extras.append(extra)
        if len(self.extras) <= self.fail_first:
            return 3
        assert jitter == 1e-8 and noise.size == noise_len
        X, R = c["X"], c["R"]
        ls = theta[1:]
        K = O.kern_K(KINDS[kind], X, None, theta[0], ls, bool(ard))
        res = O.exact_inference(K + np.eye(X.shape[0]) * extra, R, noise if noise_len > 1 else noise[0])
        res["K"] = K
        dv, dl = O.update_gradients_full(KINDS[kind], res["dL_dK"], X, None, theta[0], ls, bool(ard))
        _arr(dtheta, theta.size)[:] = np.r_[dv, dl]
        self._finish(c, res, out, alpha, diag)
        return 0

    def mi355gp_inference_given_K(self, ctx, K, noise, noise_len, jitter, extra, out, alpha, diag, ms):
        c = self.ctxs[ctx.value]
        self.calls.append("inference_given_K")
        res = O.exact_inference(K + np.eye(K.shape[0]) * extra, c["R"], noise if noise_len > 1 else noise[0])
        res["K"] = K
        self._finish(c, res, out, alpha, diag)
        return 0

    def mi355gp_fetch(self, ctx, which, out, fortran):
        self.calls.append("fetch%d" % which)
        r = self.ctxs[ctx.value]["last"]
        M = {0: r["L"], 1: r["Wi"], 2: r["dL_dK"], 3: r["K"]}[which]
        out[:] = M.T if fortran else M
        return 0

    def mi355gp_kern_K(self, device, kind, ard, theta, X, N, X2p, M, D, out):
        X2 = None if not X2p else np.ctypeslib.as_array(X2p, shape=(M, D))
        out[:] = O.kern_K(KINDS[kind], X, X2, theta[0], theta[1:], bool(ard))
        return 0

    def mi355gp_update_gradients_full(self, device, kind, ard, theta, G, X, N, X2p, M, D, g):
        X2 = None if not X2p else np.ctypeslib.as_array(X2p, shape=(M, D))
        dv, dl = O.update_gradients_full(KINDS[kind], G, X, X2, theta[0], theta[1:], bool(ard))
        g[:] = np.r_[dv, dl]
        self.calls.append("update_gradients_full")
        return 0

    # ---- sparse (VarDTC) entry points, computed by oracle/sparse_oracle.py ----------------------------------------------
    def mi355gp_sparse_create(self, device, ref):
        return self.mi355gp_create(device, ref)

    def mi355gp_sparse_set_data(self, ctx, X, N, D, Y, Dy):
        assert X.shape == (N, D) and Y.shape == (N, Dy)
        self.ctxs[ctx.value].update(X=X.copy(), R=Y.copy())
        self.calls.append("sparse_set_data")
        return 0

    def mi355gp_vardtc_inference(self, ctx, kind, ard, theta, Z, M, noise_var, extra, out, dtheta, dZ, wv, ms):
        from oracle import sparse_oracle as SO
        c = self.ctxs[ctx.value]
        self.calls.append("vardtc_inference")
        self.extras.append(extra)
        if len(self.extras) <= self.fail_first:
            return 2
        assert Z.shape == (M, c["X"].shape[1]) and extra == 0.0 or self.fail_first
        r = SO.vardtc(KINDS[kind], c["X"], Z, c["R"], theta[0], theta[1:], bool(ard), noise_var)
        out[:] = 0.0
        out[0], out[1], out[5] = r["lml"], r["dnoise"], 1.0 / max(noise_var, 1e-8)
        _arr(dtheta, theta.size)[:] = r["dtheta"]
        _arr(dZ, Z.size)[:] = r["dZ"].ravel()
        _arr(wv, M * c["R"].shape[1])[:] = r["woodbury_vector"].ravel()
        c["sparse"] = r
        return 0

    def mi355gp_sparse_fetch(self, ctx, which, out):
        r = self.ctxs[ctx.value]["sparse"]
        out[:] = {0: r["dL_dKmm"], 1: r["woodbury_inv"], 2: r["Lm"], 3: r["Kmm"]}[which]
        self.calls.append("sparse_fetch%d" % which)
        return 0

    def mi355gp_sparse_fetch_dLdKnm(self, ctx, row0, nrows, out):
        out[:] = self.ctxs[ctx.value]["sparse"]["dL_dKnm"][row0:row0 + nrows]
        self.calls.append("sparse_fetch_dLdKnm")
        return 0


@pytest.fixture(scope="module")
def env():
    ns = ref_loader.load()
    import gpy_mi355x
    gpy = type("G", (), {})()
    gpy.RBF, gpy.Matern52, gpy.Matern32, gpy.Exponential = ns.RBF, ns.Matern52, ns.Matern32, ns.Exponential
    gpy.PosteriorExact = ns.PosteriorExact
    gpy.LatentFunctionInference = sys.modules["GPy.inference.latent_function_inference"].LatentFunctionInference
    return ns, gpy_mi355x, gpy


def _reference_iteration(ns, kind, X, Y, var, ls, ARD, noise):
    return ref_loader.run_iteration(ns, kind, X, Y, var, ls if ARD else float(ls[0]), ARD, noise)


@pytest.mark.parametrize("kind,cls,ARD", [("rbf", "RBF", True), ("matern52", "Matern52", True), ("matern32", "Matern32", False),
                                          ("exponential", "Exponential", False)])
def test_stub_runs_the_reference_loop_body(env, kind, cls, ARD):
    """core/gp.py:278-280 with the stub's inference method + kernel: same LML, alpha, gradients (installed into the
    reference's own Param objects) as the reference's classes, and nothing N x N is fetched along the way."""
    ns, stub, gpy = env
    mock = OracleMockLib()
    C = stub.make_classes(mock, gpy)
    X, Y = O.synthetic(180, 3, seed=2, Dy=2)
    var, ls, noise = O.default_theta(3, ARD)
    ref = _reference_iteration(ns, kind, X, Y, var, ls, ARD, noise)
    k = getattr(C, cls)(3, variance=var, lengthscale=ls if ARD else float(ls[0]), ARD=ARD)
    lik = ns.Gaussian(variance=noise)
    inf = C.ExactGaussianInference()
    post, lml, gd = inf.inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    assert abs(lml - ref["lml"]) <= 1e-12 * abs(ref["lml"])
    assert np.abs(np.asarray(post.woodbury_vector) - ref["alpha"]).max() <= 1e-12 * np.abs(ref["alpha"]).max()
    assert np.allclose(np.asarray(k.variance.gradient), ref["dvar"], rtol=1e-10)
    assert np.allclose(np.asarray(k.lengthscale.gradient), ref["dlen"], rtol=1e-10)
    assert np.allclose(np.asarray(lik.variance.gradient), ref["dnoise"], rtol=1e-10)
    assert not any(c.startswith("fetch") for c in mock.calls) and "update_gradients_full" not in mock.calls
    # lazy members materialise on demand, in the layouts GPy expects
    Lw = np.asarray(post.woodbury_chol)
    assert Lw.flags.f_contiguous and np.abs(Lw - ref["L"]).max() <= 1e-12
    assert np.abs(np.asarray(gd["dL_dK"]) - ref["dL_dK"]).max() <= 1e-12
    # second iteration on the same data: no re-upload
    n_up = mock.calls.count("set_data")
    inf.inference(k, X, lik, Y)
    assert mock.calls.count("set_data") == n_up
    # K() through the device entry point (wrapped by the reference's slicing metaclass)
    assert np.abs(k.K(X) - ref["K"]).max() <= 1e-13
    # a foreign dL_dK goes to the generic device reduction
    k.update_gradients_full(ref["dL_dK"], X)
    assert "update_gradients_full" in mock.calls and np.allclose(np.asarray(k.lengthscale.gradient), ref["dlen"], rtol=1e-10)


def test_stub_jitter_ladder_and_error_messages(env):
    ns, stub, gpy = env
    X, Y = O.synthetic(40, 2, seed=1)
    mock = OracleMockLib(fail_first=3)
    C = stub.make_classes(mock, gpy)
    k = C.RBF(2, variance=2.0, lengthscale=1.0)
    C.ExactGaussianInference().inference(k, X, ns.Gaussian(variance=0.5), Y)
    base = (2.0 + 0.5 + 1e-8) * 1e-6                       # mean(diag Ky) * 1e-6, then x10 per try (util/linalg.py:66-72)
    assert np.allclose(mock.extras, [0.0, base, 10 * base, 100 * base], rtol=1e-12)
    mock = OracleMockLib(fail_first=99)
    C = stub.make_classes(mock, gpy)
    with pytest.raises(np.linalg.LinAlgError, match="even with jitter"):
        C.ExactGaussianInference().inference(C.RBF(2), X, ns.Gaussian(variance=0.5), Y)
    assert len(mock.extras) == 6                           # plain try + maxtries = 5


def test_stub_given_K_heteroscedastic_variance_and_Z_tilde(env):
    """the `K=`, vector `variance=` and `Z_tilde=` arguments (reference test_inference.py:118-133)"""
    ns, stub, gpy = env
    mock = OracleMockLib()
    C = stub.make_classes(mock, gpy)
    X, Y = O.synthetic(60, 2, seed=4)
    K = O.kern_K("matern32", X, None, 1.1, 0.8, False)
    nv = 0.05 + 0.1 * np.random.default_rng(0).random(60)
    ref = O.exact_inference(K, Y, nv, Z_tilde=1.25)
    post, lml, gd = C.ExactGaussianInference().inference(None, X, ns.Gaussian(variance=1.0), Y, K=K, variance=nv,
                                                         Z_tilde=1.25)
    assert "inference_given_K" in mock.calls and abs(lml - ref["lml"]) <= 1e-12 * abs(ref["lml"])
    assert abs(gd["dL_dthetaL"] - ref["dL_dnoise"]) <= 1e-10 * abs(ref["dL_dnoise"])


@pytest.mark.parametrize("kind,cls,ARD,Dy", [("rbf", "RBF", True, 1), ("matern52", "Matern52", False, 2)])
def test_sparse_stub_feeds_the_reference_update_gradients(env, kind, cls, ARD, Dy):
    """integration `make_sparse_classes(...).VarDTC` against the reference's own `VarDTC` + the body of
    `SparseGP._update_gradients` (core/sparse_gp.py:108-118) run with the REFERENCE's kernel on the stub's grad_dict: same
    posterior, LML and gradients; the fused device reductions agree with what the reference assembles from the three matrices."""
    import importlib
    from oracle.sparse_oracle import synthetic_Z
    ns, stub, gpy = env
    gpy.Posterior = importlib.import_module("GPy.inference.latent_function_inference.posterior").Posterior
    vd = importlib.import_module("GPy.inference.latent_function_inference.var_dtc")
    mock = OracleMockLib()
    S = stub.make_sparse_classes(mock, gpy)
    C = stub.make_classes(mock, gpy)
    X, Y = O.synthetic(260, 3, seed=5, Dy=Dy)
    Z = synthetic_Z(X, 19, 5)
    var, ls, noise = O.default_theta(3, ARD)
    ls_arg = ls if ARD else float(ls[0])

    def update_gradients(k, gd, Zv):                      # core/sparse_gp.py:108-118, verbatim order
        k.update_gradients_diag(gd["dL_dKdiag"], X)
        kg = np.r_[np.asarray(k.variance.gradient, float).ravel(), np.asarray(k.lengthscale.gradient, float).ravel()].copy()
        k.update_gradients_full(np.asarray(gd["dL_dKnm"]), X, Zv)
        kg += np.r_[np.asarray(k.variance.gradient, float).ravel(), np.asarray(k.lengthscale.gradient, float).ravel()]
        k.update_gradients_full(np.asarray(gd["dL_dKmm"]), Zv, None)
        kg += np.r_[np.asarray(k.variance.gradient, float).ravel(), np.asarray(k.lengthscale.gradient, float).ravel()]
        dZ = k.gradients_X(np.asarray(gd["dL_dKmm"]), Zv) + k.gradients_X(np.asarray(gd["dL_dKnm"]).T, Zv, X)
        return kg, dZ

    k_ref = ref_loader.make_kernel(ns, kind, 3, var, ls_arg, ARD)
    post_r, lml_r, gd_r = vd.VarDTC().inference(k_ref, X, Z, ns.Gaussian(variance=noise), Y)
    kg_r, dZ_r = update_gradients(k_ref, gd_r, Z)

    k_dev = getattr(C, cls)(3, variance=var, lengthscale=ls_arg, ARD=ARD)            # device kernel class of the stub
    post, lml, gd = S.VarDTC().inference(k_dev, X, Z, ns.Gaussian(variance=noise), Y)
    assert abs(lml - float(np.asarray(lml_r).ravel()[0])) <= 1e-10 * abs(float(np.asarray(lml_r).ravel()[0]))
    assert np.abs(post.woodbury_vector - post_r.woodbury_vector).max() <= 1e-8 * np.abs(post_r.woodbury_vector).max()
    assert np.abs(post.woodbury_inv - post_r.woodbury_inv).max() <= 1e-6 * np.abs(post_r.woodbury_inv).max()
    assert abs(gd["dL_dthetaL"] - float(np.asarray(gd_r["dL_dthetaL"]).ravel()[0])) <= 1e-7 * abs(float(np.asarray(gd_r["dL_dthetaL"]).ravel()[0]))
    assert np.allclose(gd["dL_dKdiag"], gd_r["dL_dKdiag"], rtol=1e-12)
    # the reference's own gradient assembly, with the REFERENCE kernel, on the stub's grad_dict (dL_dKnm materialised in blocks)
    k_ref2 = ref_loader.make_kernel(ns, kind, 3, var, ls_arg, ARD)
    kg, dZ = update_gradients(k_ref2, gd, Z)
    assert "sparse_fetch_dLdKnm" in mock.calls
    assert np.abs(kg - kg_r).max() <= 1e-6 * np.abs(kg_r).max() and np.abs(dZ - dZ_r).max() <= 1e-6 * np.abs(dZ_r).max()
    # ... and the fused reductions that travel with it are the same numbers
    assert np.abs(gd["fused"]["dtheta"] - kg_r).max() <= 1e-6 * np.abs(kg_r).max()
    assert np.abs(gd["fused"]["dZ"] - dZ_r).max() <= 1e-6 * np.abs(dZ_r).max()
    # prediction through the reference's Posterior object (posterior.py:198-262)
    Xs = np.random.default_rng(1).standard_normal((11, 3))
    mu, v = post._raw_predict(k_ref2, Xs, Z, full_cov=False)
    mu_r, v_r = post_r._raw_predict(k_ref, Xs, Z, full_cov=False)
    assert np.abs(mu - mu_r).max() <= 1e-8 and np.abs(v - v_r).max() <= 1e-8

"""Reference-side binding of libmi355gp.so -- the file a GPy maintainer adds (e.g. as `GPy/util/mi355gp.py`).

GPy has no FFI / plugin ABI: its exact-GP hot path is three Python call signatures.  This module binds
the C-ABI of `include/mi355gp.h` with ctypes and derives, FROM GPY'S OWN CLASSES, drop-in replacements for

    ExactGaussianInference.inference            GPy/inference/latent_function_inference/exact_gaussian_inference.py:37-74
    Stationary.K / Kdiag / update_gradients_full  GPy/kern/src/stationary.py:105-115,170-173,193-213
    PosteriorExact._raw_predict                 GPy/inference/latent_function_inference/posterior.py:273-302

    import GPy
    from gpy_mi355x import bind, make_classes
    C = make_classes(bind("/path/to/libmi355gp.so"))            # imports GPy's base classes itself
    m = GPy.core.GP(X, Y, C.RBF(D, ARD=True), GPy.likelihoods.Gaussian(),
                    inference_method=C.ExactGaussianInference())

`gpy_amd/` packages the same binding with paramz-free stand-ins of the GPy classes so that it can be used and tested
without GPy installed.  tests/test_integration_stub.py executes THIS file against the reference's unmodified classes
(through oracle/ref_loader.py) with the library replaced by an oracle-backed mock: that proves the reference-side glue
(argument order, gradient installation, lazy N x N fetches, the jitter ladder) independently of the device code.
"""
import ctypes
import types

import numpy as np

KIND = {"rbf": 0, "RBF": 0, "ExpQuad": 0, "Mat52": 1, "Matern52": 1, "Mat32": 2, "Matern32": 2, "Exponential": 3}
FETCH_L, FETCH_KINV, FETCH_DLDK, FETCH_K = 0, 1, 2, 3
_p = ctypes.POINTER(ctypes.c_double)


def bind(path="libmi355gp.so"):
    """ctypes.CDLL with the argument types of include/mi355gp.h (only the entry points this binding uses)."""
    from numpy.ctypeslib import ndpointer
    dp = ndpointer(np.float64, flags="C_CONTIGUOUS")
    L = ctypes.CDLL(path)
    ci, i64, cd, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    L.mi355gp_last_error.restype = ctypes.c_char_p
    L.mi355gp_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.mi355gp_destroy.argtypes = [vp]
    L.mi355gp_set_data.argtypes = [vp, dp, i64, ci, dp, ci]
    L.mi355gp_exact_inference.argtypes = [vp, ci, ci, dp, dp, i64, cd, cd, dp, _p, _p, _p, _p]
    L.mi355gp_inference_given_K.argtypes = [vp, dp, dp, i64, cd, cd, dp, _p, _p, _p]
    L.mi355gp_fetch.argtypes = [vp, ci, dp, ci]
    L.mi355gp_kern_K.argtypes = [ci, ci, ci, dp, dp, i64, _p, i64, ci, dp]
    L.mi355gp_update_gradients_full.argtypes = [ci, ci, ci, dp, dp, dp, i64, _p, i64, ci, dp]
    L.mi355gp_predict.argtypes = [vp, ci, ci, dp, dp, i64, _p, _p, ci]
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


def make_classes(L, gpy=None):
    """Builds the drop-in classes on top of GPy's own.  `gpy`: namespace with RBF, Matern52, Matern32, Exponential,
    PosteriorExact, LatentFunctionInference (default: imported from an installed GPy)."""
    if gpy is None:
        import GPy
        from GPy.inference.latent_function_inference import LatentFunctionInference
        from GPy.inference.latent_function_inference.posterior import PosteriorExact
        gpy = types.SimpleNamespace(RBF=GPy.kern.RBF, Matern52=GPy.kern.Matern52, Matern32=GPy.kern.Matern32,
                                    Exponential=GPy.kern.Exponential, PosteriorExact=PosteriorExact,
                                    LatentFunctionInference=LatentFunctionInference)
    LinAlgError = np.linalg.LinAlgError

    def check(rc, what):
        if rc < 0:
            raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, L.mi355gp_last_error().decode()))
        return rc

    class LazyFetch(object):
        """N x N result resident in HBM; crosses PCIe only if a foreign consumer asks for it (np.asarray)."""
        __array_priority__ = 100.0
        ndim = 2

        def __init__(self, owner, which, n, token, fortran=False, kern=None, dtheta=None):
            self._owner, self._which, self._n, self._token, self._fortran = owner, which, n, token, fortran
            self._kern, self.dtheta, self._host = kern, dtheta, None
            self.shape = (n, n)

        def __array__(self, dtype=None, copy=None):
            if self._host is None:
                if self._owner._token != self._token:
                    raise RuntimeError("device-resident result overwritten by a later inference call")
                out = np.empty((self._n, self._n))
                check(L.mi355gp_fetch(self._owner._ctx, self._which, out, int(self._fortran)), "mi355gp_fetch")
                self._host = out.T if self._fortran else out
            return self._host if dtype is None else self._host.astype(dtype, copy=False)

        def __getitem__(self, idx):
            return self.__array__()[idx]

    def theta_of(kern):
        return _f64(np.r_[float(np.asarray(kern.variance).ravel()[0]), np.asarray(kern.lengthscale, dtype=float).ravel()])

    class ExactGaussianInference(gpy.LatentFunctionInference):
        """`inference_method=` object for GPy.core.GP (core/gp.py:38,97-103)."""

        def __init__(self, device=0, maxtries=5):
            self.device, self.maxtries = device, maxtries
            self._ctx, self._X, self._R, self._token = None, None, None, 0

        def _ctx_for(self, X, R):
            if self._ctx is None:
                self._ctx = ctypes.c_void_p()
                check(L.mi355gp_create(self.device, ctypes.byref(self._ctx)), "mi355gp_create")
            if self._X is None or self._X.shape != X.shape or self._R.shape != R.shape or \
                    not (np.array_equal(self._X, X) and np.array_equal(self._R, R)):
                check(L.mi355gp_set_data(self._ctx, X, X.shape[0], X.shape[1], R, R.shape[1]), "mi355gp_set_data")
                self._X, self._R = X.copy(), R.copy()
            return self._ctx

        def _ladder(self, attempt, diagA):
            """jitchol's ladder (util/linalg.py:56-75) on the device's LAPACK-style `info`"""
            if attempt(0.0) == 0:
                return
            if np.any(diagA <= 0.0):
                raise LinAlgError("not pd: non-positive diagonal elements")
            jitter, tries = float(np.mean(diagA)) * 1e-6, 1
            while tries <= self.maxtries and np.isfinite(jitter):
                if attempt(jitter) == 0:
                    return
                jitter *= 10
                tries += 1
            raise LinAlgError("not positive definite, even with jitter.")

        def inference(self, kern, X, likelihood, Y, mean_function=None, Y_metadata=None, K=None, variance=None,
                      Z_tilde=None):
            m = 0 if mean_function is None else mean_function.f(X)
            if variance is None:
                variance = likelihood.gaussian_variance(Y_metadata)
            noise = _f64(np.atleast_1d(np.asarray(variance, dtype=float)).ravel())
            R = _f64(Y - m)
            n = R.shape[0]
            fused = K is None and getattr(kern, "_mi355gp_kind", None) is not None
            Xd = _f64(kern._slice_X(X) if (fused and hasattr(kern, "_slice_X")) else np.asarray(X))
            ctx = self._ctx_for(Xd, R)
            self._token += 1
            out, alpha, diag = np.zeros(8), np.empty_like(R), np.empty(n)
            if fused:
                theta = theta_of(kern)
                dtheta = np.empty(theta.size)
                self._ladder(lambda extra: check(L.mi355gp_exact_inference(
                    ctx, kern._mi355gp_kind, int(kern.ARD), theta, noise, noise.size, 1e-8, extra, out, _ptr(alpha),
                    _ptr(dtheta), _ptr(diag), None), "mi355gp_exact_inference"), theta[0] + noise + 1e-8)
                K_view = LazyFetch(self, FETCH_K, n, self._token)
                dL_dK = LazyFetch(self, FETCH_DLDK, n, self._token, kern=kern, dtheta=dtheta)
            else:
                Kh = _f64(kern.K(X) if K is None else K)
                self._ladder(lambda extra: check(L.mi355gp_inference_given_K(
                    ctx, Kh, noise, noise.size, 1e-8, extra, out, _ptr(alpha), _ptr(diag), None),
                    "mi355gp_inference_given_K"), np.diag(Kh) + noise + 1e-8)
                K_view, dL_dK = Kh, LazyFetch(self, FETCH_DLDK, n, self._token)
            lml = out[0] + (0.0 if Z_tilde is None else Z_tilde)
            dL_dthetaL = likelihood.exact_inference_gradients(diag, Y_metadata)
            post = gpy.PosteriorExact(woodbury_chol=LazyFetch(self, FETCH_L, n, self._token, fortran=True),
                                      woodbury_vector=alpha, K=K_view)
            return post, lml, {"dL_dK": dL_dK, "dL_dthetaL": dL_dthetaL, "dL_dm": alpha}

    def device_kernel(base, kind):
        class DeviceKernel(base):
            _mi355gp_kind = kind

            def K(self, X, X2=None):
                X = _f64(X)
                X2c = None if X2 is None else _f64(X2)
                out = np.empty((X.shape[0], X.shape[0] if X2c is None else X2c.shape[0]))
                check(L.mi355gp_kern_K(0, kind, int(self.ARD), theta_of(self), X, X.shape[0], _ptr(X2c), out.shape[1],
                                       X.shape[1], out), "mi355gp_kern_K")
                return out

            def update_gradients_full(self, dL_dK, X, X2=None):
                if isinstance(dL_dK, LazyFetch) and X2 is None and dL_dK._kern is self and dL_dK.dtheta is not None:
                    g = dL_dK.dtheta                     # reduced on the device in the same pass as the inference
                else:                                    # foreign dL_dK (EP, sparse GP, ...): generic device reduction
                    X = _f64(X)
                    X2c = None if X2 is None else _f64(X2)
                    G = _f64(np.asarray(dL_dK))
                    g = np.empty(1 + np.asarray(self.lengthscale).size)
                    check(L.mi355gp_update_gradients_full(0, kind, int(self.ARD), theta_of(self), G, X, X.shape[0],
                                                          _ptr(X2c), 0 if X2c is None else X2c.shape[0], X.shape[1], g),
                          "mi355gp_update_gradients_full")
                self.variance.gradient = g[0]
                self.lengthscale.gradient = g[1:] if self.ARD else g[1]
        DeviceKernel.__name__ = base.__name__ + "_MI355X"
        return DeviceKernel

    return types.SimpleNamespace(LazyFetch=LazyFetch, ExactGaussianInference=ExactGaussianInference,
                                 RBF=device_kernel(gpy.RBF, 0), Matern52=device_kernel(gpy.Matern52, 1),
                                 Matern32=device_kernel(gpy.Matern32, 2), Exponential=device_kernel(gpy.Exponential, 3))


def bind_sparse(L):
    """argument types of the sparse (VarDTC) entry points of include/mi355gp.h on an already bound library"""
    from numpy.ctypeslib import ndpointer
    dp = ndpointer(np.float64, flags="C_CONTIGUOUS")
    ci, i64, cd, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    L.mi355gp_sparse_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.mi355gp_sparse_destroy.argtypes = [vp]
    L.mi355gp_sparse_set_data.argtypes = [vp, dp, i64, ci, dp, ci]
    L.mi355gp_vardtc_inference.argtypes = [vp, ci, ci, dp, dp, i64, cd, cd, dp, _p, _p, _p, _p]
    L.mi355gp_sparse_fetch.argtypes = [vp, ci, dp]
    L.mi355gp_sparse_fetch_dLdKnm.argtypes = [vp, i64, i64, dp]
    return L


SP_DLDKMM, SP_WOODBURY_INV, SP_LM, SP_KMM = 0, 1, 2, 3


def make_sparse_classes(L, gpy=None):
    """`inference_method=` object for GPy.core.SparseGP (core/sparse_gp.py:49-56,76-119): a `VarDTC` whose `inference` runs on
    the device and hands the reference's own `SparseGP._update_gradients` a `grad_dict` with the reference's keys.
    `gpy`: namespace with LatentFunctionInference and Posterior (default: imported from an installed GPy)."""
    if gpy is None:
        from GPy.inference.latent_function_inference import LatentFunctionInference
        from GPy.inference.latent_function_inference.posterior import Posterior
        gpy = types.SimpleNamespace(LatentFunctionInference=LatentFunctionInference, Posterior=Posterior)
    LinAlgError = np.linalg.LinAlgError

    def check(rc, what):
        if rc < 0:
            raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, L.mi355gp_last_error().decode()))
        return rc

    class LazyNM(object):
        """dL_dKnm (N x M): never resident as a whole on the device (3.3 GB at N=200000, M=2048); a consumer that needs the
        array (a foreign kernel's update_gradients_full / gradients_X, core/sparse_gp.py:112-118) gets it assembled from row
        blocks, the device-capable kernels of this module take `fused` instead."""
        __array_priority__ = 100.0
        ndim = 2

        def __init__(self, owner, N, M, token, block=8192):
            self._owner, self._token, self._block, self._host = owner, token, block, None
            self.shape = (N, M)

        def __array__(self, dtype=None, copy=None):
            if self._host is None:
                if self._owner._token != self._token:
                    raise RuntimeError("device-resident result overwritten by a later inference call")
                N, M = self.shape
                out = np.empty((N, M))
                for r0 in range(0, N, self._block):
                    nr = min(self._block, N - r0)
                    blk = np.empty((nr, M))
                    check(L.mi355gp_sparse_fetch_dLdKnm(self._owner._ctx, r0, nr, blk), "mi355gp_sparse_fetch_dLdKnm")
                    out[r0:r0 + nr] = blk
                self._host = out
            return self._host if dtype is None else self._host.astype(dtype, copy=False)

        @property
        def T(self):
            return self.__array__().T

    class VarDTC(gpy.LatentFunctionInference):
        const_jitter = 1e-8

        def __init__(self, device=0, maxtries=5):
            self.device, self.maxtries = device, maxtries
            self._ctx, self._X, self._Y, self._token = None, None, None, 0

        def _ctx_for(self, X, Y):
            if self._ctx is None:
                self._ctx = ctypes.c_void_p()
                check(L.mi355gp_sparse_create(self.device, ctypes.byref(self._ctx)), "mi355gp_sparse_create")
            if self._X is None or self._X.shape != X.shape or self._Y.shape != Y.shape or \
                    not (np.array_equal(self._X, X) and np.array_equal(self._Y, Y)):
                check(L.mi355gp_sparse_set_data(self._ctx, X, X.shape[0], X.shape[1], Y, Y.shape[1]), "mi355gp_sparse_set_data")
                self._X, self._Y = X.copy(), Y.copy()
            return self._ctx

        def _fetch(self, which, M):
            out = np.empty((M, M))
            check(L.mi355gp_sparse_fetch(self._ctx, which, out), "mi355gp_sparse_fetch")
            return out

        def inference(self, kern, X, Z, likelihood, Y, Y_metadata=None, mean_function=None, precision=None, Lm=None,
                      dL_dKmm=None, psi0=None, psi1=None, psi2=None, Z_tilde=None):
            """(reference `var_dtc.py:66-215`) for a device kernel of this module, certain inputs, a homoscedastic Gaussian
            likelihood and no mean function -- anything else belongs to the reference's own VarDTC."""
            kind = getattr(kern, "_mi355gp_kind", None)
            if kind is None or mean_function is not None or precision is not None or psi1 is not None:
                raise NotImplementedError("mi355gp VarDTC: device stationary kernel, certain inputs, no mean function")
            noise = float(np.asarray(likelihood.gaussian_variance(Y_metadata)).ravel()[0])
            Xd, Zd, Yd = _f64(kern._slice_X(X) if hasattr(kern, "_slice_X") else X), _f64(Z), _f64(Y)
            ctx = self._ctx_for(Xd, Yd)
            self._token += 1
            N, M, Dy = Xd.shape[0], Zd.shape[0], Yd.shape[1]
            theta = _f64(np.r_[float(np.asarray(kern.variance).ravel()[0]), np.asarray(kern.lengthscale, dtype=float).ravel()])
            out, dtheta, dZ, wv = np.zeros(8), np.empty(theta.size), np.empty((M, Zd.shape[1])), np.empty((M, Dy))

            def attempt(extra):
                return check(L.mi355gp_vardtc_inference(ctx, kind, int(kern.ARD), theta, Zd, M, noise, extra, out, _ptr(dtheta),
                                                        _ptr(dZ), _ptr(wv), None), "mi355gp_vardtc_inference")
            if attempt(0.0) != 0:                                   # jitchol's ladder on Kmm / B (util/linalg.py:56-75)
                jitter, tries, ok = theta[0] * 1e-6, 1, False
                while tries <= self.maxtries and not ok:
                    ok = attempt(jitter) == 0
                    jitter *= 10
                    tries += 1
                if not ok:
                    raise LinAlgError("not positive definite, even with jitter.")
            beta = out[5]
            post = gpy.Posterior(woodbury_inv=self._fetch(SP_WOODBURY_INV, M), woodbury_vector=wv,
                                 K=self._fetch(SP_KMM, M), mean=None, cov=None, K_chol=self._fetch(SP_LM, M))
            grad_dict = {"dL_dKmm": self._fetch(SP_DLDKMM, M),
                         "dL_dKdiag": -0.5 * Dy * beta * np.ones(N),                 # var_dtc.py:218 (dL_dpsi0)
                         "dL_dKnm": LazyNM(self, N, M, self._token),
                         "dL_dthetaL": out[1],
                         # what SparseGP._update_gradients would assemble from the three matrices above
                         # (core/sparse_gp.py:108-118), reduced on the device in the same two passes:
                         "fused": {"kern": kern, "dtheta": dtheta, "dZ": dZ}}
            return post, out[0] + (0.0 if Z_tilde is None else Z_tilde), grad_dict

    return types.SimpleNamespace(VarDTC=VarDTC, LazyNM=LazyNM)

"""Sparse GP regression (VarDTC) backed by libmi355gp.so -- drop-in for the hot path of
`GPy.inference.latent_function_inference.VarDTC` + `GPy.core.SparseGP` / `GPy.models.SparseGPRegression`
(reference `var_dtc.py:66-215`, `core/sparse_gp.py:76-119`, `models/sparse_gp_regression.py:20-60`) for certain inputs
and a homoscedastic Gaussian likelihood (BASELINE config 5).

`VarDTC.inference(kern, X, Z, likelihood, Y, ...)` returns `(Posterior, log_marginal, grad_dict)` with the reference's
keys, for gpy_amd's stationary kernels and sums (`Add`) of stationary / White / Bias parts, a scalar or per-point
Gaussian noise variance and an optional mean function.  The N x M matrix `dL_dKnm` is never resident: the kernel and
inducing-input gradients that `SparseGP._update_gradients` derives from it are reduced on the device in the second
streaming pass and travel in `grad_dict['fused']`; `dL_dKmm` is a lazy device view and `dL_dKnm` a lazy object that a
foreign consumer can materialise in row blocks (`mi355gp_sparse_fetch_dLdKnm`).
"""
import numpy as np

from . import _lib
from .kern import RBF, Stationary
from .lazy import ArrayIdentity, freeze
from .likelihoods import Gaussian
from .models import PredictionCallers
from .param import Param, Parameterized

LinAlgError = np.linalg.LinAlgError


class _LazyMM(object):
    """Lazy M x M result of the last VarDTC call (fetched on np.asarray)."""
    __array_priority__ = 100.0

    def __init__(self, ctx, which, M, token, owner):
        self._ctx, self._which, self._M, self._token, self._owner = ctx, which, M, token, owner
        self._host = None

    shape = property(lambda self: (self._M, self._M))
    ndim = 2
    dtype = np.dtype(np.float64)

    def fetch(self):
        if self._host is None:
            if self._owner._token != self._token:
                raise RuntimeError("device-resident result overwritten by a later inference call")
            self._host = self._ctx.fetch(self._which)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.fetch()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.fetch()[idx]

    @property
    def T(self):
        return self.fetch().T


class _LazyNM(object):
    """dL_dKnm (N x M) of the last VarDTC call.  It is never resident as a whole on the device (3.3 GB at N = 200000,
    M = 2048): gpy_amd's own model driver uses the reductions the device made from it (`grad_dict['fused']`); a FOREIGN
    consumer -- e.g. GPy's `SparseGP._update_gradients` handing it to `kern.update_gradients_full(dL_dKnm, X, Z)`
    (reference `core/sparse_gp.py:108-119`) -- gets it materialised in row blocks (`blocks()`), or entirely (`np.asarray`)."""
    __array_priority__ = 100.0
    ndim = 2
    dtype = np.dtype(np.float64)

    def __init__(self, ctx, N, M, token, owner, block=8192):
        self._ctx, self.shape, self._token, self._owner, self._block = ctx, (N, M), token, owner, block
        self._host = None

    def blocks(self, rows=None):
        """yields (row0, row1, dL_dKnm[row0:row1]) over the whole matrix"""
        if self._owner._token != self._token:
            raise RuntimeError("device-resident result overwritten by a later inference call")
        step = rows or self._block
        for r0 in range(0, self.shape[0], step):
            r1 = min(self.shape[0], r0 + step)
            yield r0, r1, self._ctx.fetch_dL_dKnm(r0, r1 - r0)

    def fetch(self):
        if self._host is None:
            out = np.empty(self.shape)
            for r0, r1, B in self.blocks():
                out[r0:r1] = B
            self._host = out
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.fetch()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.fetch()[idx]

    @property
    def T(self):
        return self.fetch().T


class SparsePosterior(object):
    """`Posterior(woodbury_inv, woodbury_vector, K=Kmm, K_chol=Lm)` of the reference (`var_dtc.py:213`,
    `posterior.py:21-77`); prediction follows `Posterior._raw_predict` (`posterior.py:198-262`) and runs on the device
    (C-ABI `mi355gp_sparse_predict`) while the posterior is the context's latest result."""

    def __init__(self, woodbury_inv, woodbury_vector, K, K_chol, device=None):
        self.woodbury_inv, self.woodbury_vector, self.K, self.K_chol = woodbury_inv, woodbury_vector, K, K_chol
        self._device = device            # (ctx, token, owner, kernel signature, specs builder)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_device"] = None
        for k in ("woodbury_inv", "K", "K_chol"):
            d[k] = np.asarray(d[k])
        return d

    def _raw_predict(self, kern, Xnew, pred_var, full_cov=False):
        dev = self._device
        if dev is not None and dev["owner"]._token == dev["token"] and _kernel_sig(kern) == dev["sig"]:
            Xn = kern._slice_X(np.asarray(Xnew)) if isinstance(kern, Stationary) else _lib.f64(Xnew)
            return dev["ctx"].predict(_specs(kern), Xn, full_cov=full_cov)
        Kx = kern.K(pred_var, Xnew)                                   # (M, N*): foreign kernel / stale device state
        mu = np.dot(Kx.T, self.woodbury_vector)
        Wi = np.asarray(self.woodbury_inv)
        if full_cov:
            var = kern.K(Xnew) - np.dot(Kx.T, np.dot(Wi, Kx))
        else:
            var = (kern.Kdiag(Xnew) - np.sum(np.dot(Wi.T, Kx) * Kx, 0))[:, None]
            var = np.clip(var, 1e-15, np.inf)                        # posterior.py:248
        return mu, var

    def predictive_gradients(self, kern, Xnew, pred_var=None):
        """(reference `core/gp.py:418-474` with `_predictive_variable` = Z): M is small, the M x M Woodbury inverse comes to
        the host and the reductions over the inducing points run through `kern.gradients_X`."""
        return _host_predictive_gradients(kern, Xnew, pred_var, self.woodbury_vector, self.woodbury_inv)

    def covariance_between_points(self, kern, X, X1, X2):
        """K(X1, X2) - K(X1, Z) woodbury_inv K(Z, X2) (reference `posterior.py:109-130`)"""
        return kern.K(X1, X2) - np.dot(kern.K(X1, X), np.dot(np.asarray(self.woodbury_inv), kern.K(X, X2)))


def _host_predictive_gradients(kern, Xnew, pred_var, woodbury_vector, woodbury_inv):
    """`GP.predictive_gradients` (reference `core/gp.py:440-474`, woodbury_inv.ndim == 2) composed from `kern.gradients_X`
    (device reductions) and the M x M / N x N Woodbury inverse."""
    Xnew = np.asarray(Xnew, dtype=np.float64)
    wv = np.asarray(woodbury_vector)
    mean_jac = np.empty((Xnew.shape[0], Xnew.shape[1], wv.shape[1]))
    for i in range(wv.shape[1]):
        mean_jac[:, :, i] = kern.gradients_X(wv[:, i:i + 1].T * np.ones((Xnew.shape[0], 1)), Xnew, pred_var)
    var_jac = kern.gradients_X_diag(np.ones(Xnew.shape[0]), Xnew)
    a2 = -2.0 * np.dot(kern.K(Xnew, pred_var), np.asarray(woodbury_inv))
    return mean_jac, var_jac + kern.gradients_X(a2, Xnew, pred_var)


def _specs(kern):
    """[(kind, ARD, theta, active_dims, term)] of a stationary kernel (its own column slicing applied to X on upload) or
    of an `Add` of stationary / White / Bias parts (active_dims index the model's X)"""
    if isinstance(kern, Stationary):
        return [(kern.kind, kern.ARD, kern._theta(), None, 0)]
    return kern.part_specs()


def _kernel_sig(kern):
    from .lazy import kernel_signature
    return kernel_signature(kern)


class VarDTC(object):
    const_jitter = 1e-8

    def __init__(self, device=0, maxtries=5):
        self.device, self.maxtries = device, maxtries
        self._ctx = None
        self._X = self._Y = None
        self._token = 0
        self.last_stage_ms = None
        self.collect_stage_ms = False

    def on_optimization_start(self):
        pass

    def on_optimization_end(self):
        pass

    def to_dict(self):
        return {"class": "GPy.inference.latent_function_inference.var_dtc.VarDTC"}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_ctx"] = None
        d["_X"] = d["_Y"] = None
        return d

    def _ensure(self, X, Y):
        if self._ctx is None:
            self._ctx = _lib.SparseContext(self.device)
        # frozen arrays (the model driver's X / Y) are recognised by identity, anything else by a full comparison
        if self._X is None or not (self._X.matches(X) and self._Y.matches(Y)):
            self._ctx.set_data(X, Y)
            self._X, self._Y = ArrayIdentity(X), ArrayIdentity(Y)

    def inference(self, kern, X, Z, likelihood, Y, Y_metadata=None, mean_function=None, precision=None, Lm=None,
                  dL_dKmm=None, psi0=None, psi1=None, psi2=None, Z_tilde=None):
        from .kern import Add, Prod, White
        # stationary kernels, products (Prod, reference `prod.py:58-99`) of stationary / Bias factors, and sums (Add) of those
        # and of White / Bias parts
        def _prod_ok(k):
            return isinstance(k, Prod) and not any(isinstance(f, (White, Add, Prod)) for f in k.parts)
        ok = isinstance(kern, Stationary) or _prod_ok(kern) or (
            isinstance(kern, Add) and all(_prod_ok(p) if isinstance(p, Prod) else not isinstance(p, Add) for p in kern.parts))
        if not ok:
            raise NotImplementedError("the MI355X sparse path covers gpy_amd's stationary kernels, products of stationary / "
                                      "Bias factors and sums (Add) of those and of White / Bias parts")
        if any(a is not None for a in (Lm, dL_dKmm, psi0, psi1, psi2)):
            raise NotImplementedError("precomputed statistics are not accepted by the MI355X sparse path")
        Y = np.asarray(Y, dtype=np.float64)
        if precision is None:                                          # var_dtc.py:78-80
            noise = np.atleast_1d(np.asarray(likelihood.gaussian_variance(Y_metadata), dtype=np.float64)).ravel()
        else:
            noise = 1.0 / np.atleast_1d(np.asarray(precision, dtype=np.float64)).ravel()
        het = noise.size > 1
        if het and mean_function is not None:                          # var_dtc.py:85-86
            raise ValueError("Mean function not implemented with uncertain inputs or heteroscedasticity")
        m = 0 if mean_function is None else mean_function.f(X)
        single = isinstance(kern, Stationary)
        Xs = kern._slice_X(X) if single else _lib.f64(X)
        Zs = kern._slice_X(np.asarray(Z)) if single else _lib.f64(Z)
        R = _lib.f64(Y - m)
        self._ensure(Xs, R)
        specs = _specs(kern)
        kdiag = float(kern.variance.values[0]) if single else kern.diag_variance()
        extra, tries, info = 0.0, 0, 1
        while True:                                   # jitchol's ladder (util/linalg.py:56-75) for Kmm / B
            info, r = self._ctx.vardtc_sum(specs, Zs, noise, extra_jitter=extra, want_dL_dm=mean_function is not None,
                                           want_stage_ms=self.collect_stage_ms)
            if info == 0:
                break
            if tries >= self.maxtries:
                raise LinAlgError("not positive definite, even with jitter.")
            extra = kdiag * 1e-6 * 10 ** tries
            tries += 1
        self._token += 1
        self.last_stage_ms = r.get("stage_ms")
        M, N = Zs.shape[0], Xs.shape[0]
        lml = r["lml"] + (0.0 if Z_tilde is None else Z_tilde)
        C = _lib.SparseContext
        post = SparsePosterior(woodbury_inv=_LazyMM(self._ctx, C.FETCH_WOODBURY_INV, M, self._token, self),
                               woodbury_vector=r["woodbury_vector"],
                               K=_LazyMM(self._ctx, C.FETCH_KMM, M, self._token, self),
                               K_chol=_LazyMM(self._ctx, C.FETCH_LM, M, self._token, self),
                               device={"ctx": self._ctx, "token": self._token, "owner": self, "sig": _kernel_sig(kern)})
        beta = 1.0 / np.fmax(noise, self.const_jitter)
        dL_dR = (r["dnoise"][:, None] if r["dnoise"].ndim == 1 else r["dnoise"]) if het else r["dnoise"]   # N x Dy (var_dtc.py:240-256)
        grad_dict = {"dL_dKmm": _LazyMM(self._ctx, C.FETCH_DLDKMM, M, self._token, self),
                     "dL_dKdiag": -0.5 * Y.shape[1] * (beta * np.ones(N)),            # var_dtc.py:218
                     "dL_dKnm": _LazyNM(self._ctx, N, M, self._token, self),
                     "dL_dthetaL": likelihood.exact_inference_gradients(dL_dR, Y_metadata),
                     "dL_dm": r["dL_dm"],
                     "fused": {"dtheta": r["dtheta"], "dZ": r["dZ"]}}
        return post, lml, grad_dict


class SparseGP(PredictionCallers, Parameterized):
    """Model driver: the `SparseGP.parameters_changed` sequence (reference `core/sparse_gp.py:76-119`).
    Flat parameter order follows GPy's links: [Z, kern.variance, kern.lengthscale..., noise variance]
    (`sparse_gp.py:59`: Z is linked at index 0)."""

    def __init__(self, X, Y, Z, kernel, likelihood, inference_method=None, name="sparse gp", device=0, mean_function=None,
                 Y_metadata=None):
        super(SparseGP, self).__init__(name)
        self.mean_function, self.Y_metadata = mean_function, Y_metadata
        self.X, self.Y = freeze(X), freeze(Y)        # private read-only copies (cf. `ObsAr`, reference `core/gp.py:44-60`)
        self.Y_normalized = self.Y
        self.num_data, self.input_dim = self.X.shape
        self.output_dim = self.Y.shape[1]
        self.Z = Param("inducing inputs", np.asarray(Z, dtype=np.float64), positive=False)
        self.num_inducing = self.Z.shape[0]
        self.kern, self.likelihood = kernel, likelihood
        self.inference_method = inference_method or VarDTC(device=device)
        self.link_parameter(self.Z)
        self.link_parameter(self.kern)
        self.link_parameter(self.likelihood)
        self.posterior = None
        self.parameters_changed()

    def parameters_changed(self):
        self.posterior, self._log_marginal_likelihood, self.grad_dict = self.inference_method.inference(
            self.kern, self.X, self.Z.values, self.likelihood, self.Y_normalized, Y_metadata=self.Y_metadata,
            mean_function=self.mean_function)
        self.likelihood.update_gradients(self.grad_dict["dL_dthetaL"])
        if self.mean_function is not None:                                      # sparse_gp.py:84-85
            self.mean_function.update_gradients(self.grad_dict["dL_dm"], self.X)
        fused = self.grad_dict["fused"]
        if isinstance(self.kern, Stationary):
            self.kern._install_gradients(fused["dtheta"])
        else:
            self.kern._install_fused(fused["dtheta"])
        self.Z.gradient = fused["dZ"] if fused["dZ"].shape == self.Z.shape else self._scatter_dZ(fused["dZ"])

    def _scatter_dZ(self, dZ):
        full = np.zeros(self.Z.shape)
        full[:, self.kern.active_dims] = dZ
        return full

    def log_likelihood(self):
        return self._log_marginal_likelihood

    def objective_function(self):
        return -float(self._log_marginal_likelihood)

    def objective_function_gradients(self):
        return -self.gradient

    @property
    def _predictive_variable(self):
        """(reference `core/sparse_gp.py:72-74`)"""
        return self.Z.values

    def _raw_predict(self, Xnew, full_cov=False, kern=None):
        """(reference `core/sparse_gp.py:121-160` -> `posterior.py:198-262`; same signature as `GP._raw_predict`)"""
        mu, var = self.posterior._raw_predict(self.kern if kern is None else kern, np.asarray(Xnew), self.Z.values,
                                              full_cov=full_cov)
        if self.mean_function is not None:
            mu = mu + self.mean_function.f(Xnew)
        return mu, var

    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """(reference `core/gp.py:308-365`, inherited by `SparseGP`: same argument order)"""
        mu, var = self._raw_predict(Xnew, full_cov, kern=kern)
        if include_likelihood:
            likelihood = self.likelihood if likelihood is None else likelihood
            mu, var = likelihood.predictive_values(mu, var, full_cov=full_cov,
                                                   Y_metadata=self.Y_metadata if Y_metadata is None else Y_metadata)
        return mu, var

    def optimize(self, max_iters=100, messages=False, gtol=1e-6):
        """L-BFGS-B; positive parameters (kernel, noise) in log space, Z untransformed."""
        from scipy.optimize import minimize
        pos = np.concatenate([np.full(p.size, bool(p.positive)) for p in self.flattened_parameters()])

        def to_x(p):
            return np.where(pos, np.log(np.where(pos, p, 1.0)), p)

        def from_x(x):
            return np.where(pos, np.exp(np.where(pos, x, 0.0)), x)

        def f(x):
            try:
                self.param_array = from_x(x)
            except LinAlgError:
                return 1e300, np.zeros_like(x)
            g = self.objective_function_gradients()
            return self.objective_function(), np.where(pos, g * self.param_array, g)
        res = minimize(f, to_x(self.param_array), jac=True, method="L-BFGS-B",
                       options={"maxiter": max_iters, "gtol": gtol, "disp": bool(messages)})
        self.param_array = from_x(res.x)
        return res


class SparseGPRegression(SparseGP):
    """(reference `GPy/models/sparse_gp_regression.py:20-60`): Z defaults to a random subset of X."""

    def __init__(self, X, Y, kernel=None, Z=None, num_inducing=10, noise_var=1., device=0, seed=None, mean_function=None):
        X = np.asarray(X, dtype=np.float64)
        if kernel is None:
            kernel = RBF(X.shape[1], device=device)
        if Z is None:
            i = np.random.default_rng(seed).permutation(X.shape[0])[:min(num_inducing, X.shape[0])]
            Z = X[i].copy()
        super(SparseGPRegression, self).__init__(X, Y, Z, kernel, Gaussian(variance=noise_var),
                                                 name="sparse_gp", device=device, mean_function=mean_function)

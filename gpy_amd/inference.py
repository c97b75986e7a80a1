"""`ExactGaussianInference` backed by libmi355gp.so -- drop-in for
`GPy.inference.latent_function_inference.ExactGaussianInference` (reference
`GPy/inference/latent_function_inference/exact_gaussian_inference.py:11-88`).

`inference(kern, X, likelihood, Y, mean_function=None, Y_metadata=None, K=None, variance=None, Z_tilde=None)`
returns `(Posterior, log_marginal_likelihood, {'dL_dK', 'dL_dthetaL', 'dL_dm'})` like the reference; with a
gpy_amd kernel the whole evaluation (K build, Cholesky, alpha, Ky^-1, all gradients) is one C-ABI call and the
N x N results stay in HBM behind lazy proxies.  The jitter ladder of `jitchol` (reference
`GPy/util/linalg.py:56-75`) runs here, on LAPACK-style `info` codes from the device factorisation, and raises
the same `numpy.linalg.LinAlgError` messages.
"""
import numpy as np

from . import _lib
from .kern import CombinationKernel, Stationary
from .lazy import ArrayIdentity, DeviceResult, kernel_signature
from .likelihoods import Gaussian
from .posterior import PosteriorExact, StudentTPosterior

LinAlgError = np.linalg.LinAlgError


class _DeviceState(object):
    """One uploaded (X, R) pair and the HBM buffers behind it."""

    def __init__(self, device):
        self.ctx = _lib.Context(device)
        self._idX = self._idR = None
        self.call_token = 0
        self.kern = None

    def __getstate__(self):              # the HBM buffers and the ctypes handle never travel
        return {"_idX": None, "_idR": None, "call_token": self.call_token, "kern": None, "ctx": None}

    def ensure_data(self, X, R):
        """Upload X / R unless the device already holds exactly these data.  An array the caller froze (the model drivers
        keep private read-only copies of X and Y, like GPy's `ObsAr`, reference `core/gp.py:44-60`) is recognised by
        identity in O(1); anything writable is compared element by element with a private copy (O(N D), negligible next
        to the N^3 step) -- an in-place edit of a caller-owned buffer is never missed."""
        same_x = self._idX is not None and self._idX.matches(X)
        if not same_x or self._idR is None or self._idR.value().shape != R.shape:
            self.ctx.set_data(X, R)
            self._idX, self._idR = ArrayIdentity(X), ArrayIdentity(R)
        elif not self._idR.matches(R):
            self.ctx.set_targets(R)
            self._idR = ArrayIdentity(R)

    @property
    def X(self):
        return None if self._idX is None else self._idX.value()

    @property
    def R(self):
        return None if self._idR is None else self._idR.value()

    def invalidate(self):
        """Forget what was uploaded: the next inference call uploads X and R again."""
        self._idX = self._idR = None

    def fetch(self, which, fortran_order=False):
        return self.ctx.fetch(which, fortran_order=fortran_order)

    def covariance_between_points(self, kern, X1, X2):
        if isinstance(kern, CombinationKernel):
            return self.ctx.covariance_between_points(kern.part_specs(), _lib.f64(X1), _lib.f64(X2))
        # a single kernel's active_dims were applied when X was uploaded: slice the new points the same way
        return self.ctx.covariance_between_points([(kern.kind, kern.ARD, kern._theta(), None)], kern._slice_X(X1),
                                                  kern._slice_X(X2))

    def predictive_gradients(self, kern, Xnew, want_var=True):
        """(dmu (M x D x Dy), dvar (M x D)) -- reference `core/gp.py:418-474`; raises NotImplementedError for kernel
        expressions the device entry does not take (products), which the caller then evaluates on the host."""
        from .kern import Prod
        if isinstance(kern, CombinationKernel):
            if isinstance(kern, Prod) or any(isinstance(p, Prod) for p in kern.parts):
                raise NotImplementedError("product kernels")
            return self.ctx.predictive_gradients(kern.part_specs(), _lib.f64(Xnew), want_var=want_var)
        Xs = kern._slice_X(Xnew)
        dmu, dvar = self.ctx.predictive_gradients([(kern.kind, kern.ARD, kern._theta(), None)], Xs, want_var=want_var)
        if Xs.shape[1] == np.asarray(Xnew).shape[1]:
            return dmu, dvar
        full_mu = np.zeros((Xs.shape[0], np.asarray(Xnew).shape[1], dmu.shape[2]))   # active_dims of a single kernel
        full_mu[:, kern.active_dims, :] = dmu
        full_var = None
        if dvar is not None:
            full_var = np.zeros((Xs.shape[0], np.asarray(Xnew).shape[1]))
            full_var[:, kern.active_dims] = dvar
        return full_mu, full_var

    def predict(self, kern, Xnew, full_cov=False):
        if isinstance(kern, CombinationKernel):
            return self.ctx.predict_sum(kern.part_specs(), _lib.f64(Xnew), full_cov=full_cov)
        return self.ctx.predict(kern.kind, kern.ARD, kern._theta(), kern._slice_X(Xnew), full_cov=full_cov)


class ExactGaussianInference(object):
    def __init__(self, device=0, maxtries=5):
        self.device = device
        self.maxtries = maxtries
        self._state = None
        self.last_stage_ms = None
        self.collect_stage_ms = False
        self.keep_diag = True      # N doubles per call: diag(Ky^-1) for LOO without fetching the N x N inverse
        self._last = None

    # GPy's LatentFunctionInference hooks (reference latent_function_inference/__init__.py:38-49)
    def on_optimization_start(self):
        pass

    def on_optimization_end(self):
        pass

    def to_dict(self):
        return {"class": "GPy.inference.latent_function_inference.exact_gaussian_inference.ExactGaussianInference"}

    def __getstate__(self):          # device handles never travel (cf. reference rbf.py:313-318)
        d = dict(self.__dict__)
        d["_state"] = None
        d["_last"] = None            # holds the device state of the last call
        return d

    def LOO(self, kern, X, Y, likelihood, posterior, Y_metadata=None, K=None):
        """Leave-one-out log predictive densities (reference `exact_gaussian_inference.py:76-88`):
        -(0.5 log 2pi - 0.5 log c_ii + 0.5 g_i^2 / c_ii) with g = woodbury_vector, c = Ky^-1.  diag(Ky^-1) comes from the
        device's diag(dL_dK) = 0.5 (|alpha_i|^2 - Dy c_ii); the N x N inverse is never fetched."""
        g = np.asarray(posterior.woodbury_vector)
        last = self._last
        if last is not None and last["diag_dL_dK"] is not None and last["alpha"] is g:
            c_diag = ((np.sum(g * g, axis=1) - 2.0 * last["diag_dL_dK"]) / last["Dy"])[:, None]
        else:
            c_diag = np.diag(np.asarray(posterior.woodbury_inv))[:, None]
        neg = 0.5 * np.log(2 * np.pi) - 0.5 * np.log(c_diag) + 0.5 * (g ** 2) / c_diag
        return -neg

    def _run_with_ladder(self, attempt, diagA):
        """`attempt(extra_jitter)` -> (info, result).  Mirrors jitchol: plain try, then mean(diag)*1e-6 * 10^k."""
        info, res = attempt(0.0)
        if info == 0:
            return res
        if np.any(diagA <= 0.):
            raise LinAlgError("not pd: non-positive diagonal elements")
        jitter = float(np.mean(diagA)) * 1e-6
        num_tries = 1
        while num_tries <= self.maxtries and np.isfinite(jitter):
            info, res = attempt(jitter)
            if info == 0:
                return res
            jitter *= 10
            num_tries += 1
        raise LinAlgError("not positive definite, even with jitter.")

    def inference(self, kern, X, likelihood, Y, mean_function=None, Y_metadata=None, K=None, variance=None,
                  Z_tilde=None):
        X = np.asarray(X)
        Y = np.asarray(Y, dtype=np.float64)
        if variance is None:
            variance = likelihood.gaussian_variance(Y_metadata)
        noise = np.atleast_1d(np.asarray(variance, dtype=np.float64)).ravel()
        # (reference :42-50) R = Y - m; without a mean function R IS Y (same object every iteration: O(1) data check)
        R = _lib.f64(Y) if mean_function is None else _lib.f64(Y - mean_function.f(X))
        n = X.shape[0]
        is_sum = isinstance(kern, CombinationKernel)
        fused = K is None and (isinstance(kern, Stationary) or is_sum)
        Xdev = kern._slice_X(X) if fused else _lib.f64(X)
        if self._state is None:
            self._state = _DeviceState(self.device)
        st = self._state
        st.ensure_data(Xdev, R)
        st.call_token += 1
        want_ms = self.collect_stage_ms
        # Gaussian.exact_inference_gradients == sum(diag(dL_dK)) == trace(dL_dK), which the device already reduces:
        # only likelihoods with per-row noise terms need the N-vector diag(dL_dK) shipped back.
        trace_only = isinstance(likelihood, Gaussian) and noise.size == 1
        scalar_noise_lik = trace_only

        if fused:
            if is_sum:
                specs = kern.part_specs()
                diagA = kern.diag_variance() + noise + 1e-8

                def attempt(extra):
                    return st.ctx.exact_inference_sum(specs, noise, jitter=1e-8, extra_jitter=extra, want_alpha=True,
                                                      want_diag=self.keep_diag or not scalar_noise_lik,
                                                      want_stage_ms=want_ms)
            else:
                theta = kern._theta()
                diagA = float(theta[0]) + noise + 1e-8

                def attempt(extra):
                    return st.ctx.exact_inference(kern.kind, kern.ARD, theta, noise, jitter=1e-8, extra_jitter=extra,
                                                  want_alpha=True, want_diag=self.keep_diag or not scalar_noise_lik,
                                                  want_stage_ms=want_ms)
            res = self._run_with_ladder(attempt, diagA)
            sig = kernel_signature(kern)
            K_view = DeviceResult(st, _lib.FETCH_K, n, st.call_token)
            dL_dK = DeviceResult(st, _lib.FETCH_DLDK, n, st.call_token, kernel_sig=sig, fused_dtheta=res["dtheta"])
        else:
            if K is None:
                K = kern.K(X)
            K = _lib.f64(K)
            diagA = np.diag(K) + noise + 1e-8

            def attempt(extra):
                return st.ctx.inference_given_K(K, noise, jitter=1e-8, extra_jitter=extra,
                                                want_diag=self.keep_diag or not scalar_noise_lik, want_stage_ms=want_ms)
            res = self._run_with_ladder(attempt, diagA)
            K_view = K
            dL_dK = DeviceResult(st, _lib.FETCH_DLDK, n, st.call_token)
        self.last_stage_ms = res.get("stage_ms")

        log_marginal = res["lml"]
        if Z_tilde is not None:
            log_marginal += Z_tilde
        alpha = res["alpha"]
        if trace_only:
            dL_dthetaL = res["dnoise"]
        else:
            dL_dthetaL = likelihood.exact_inference_gradients(res["diag_dL_dK"], Y_metadata)
        self._last = {"alpha": alpha, "diag_dL_dK": res.get("diag_dL_dK"), "Dy": R.shape[1], "state": st,
                      "token": st.call_token}
        post = PosteriorExact(
            woodbury_chol=DeviceResult(st, _lib.FETCH_L, n, st.call_token, fortran_order=True),
            woodbury_vector=alpha, K=K_view,
            woodbury_inv=DeviceResult(st, _lib.FETCH_KINV, n, st.call_token), state=st)
        return post, log_marginal, {"dL_dK": dL_dK, "dL_dthetaL": dL_dthetaL, "dL_dm": alpha}


class ExactStudentTInference(object):
    """Student-t PROCESS inference (reference `exact_studentt_inference.py:13-52`): `inference(kern, X, Y, nu, mean_function=None,
    K=None)` -> (posterior, log_marginal, {'dL_dK', 'dL_dnu', 'dL_dm'}).  Same device pipeline as the Gaussian case
    (C-ABI `mi355gp_exact_studentt_sum`); the kernel gradients ride on the lazy dL_dK like in `ExactGaussianInference`."""

    def __init__(self, device=0, maxtries=5):
        self.device, self.maxtries = device, maxtries
        self._state = None

    def on_optimization_start(self):
        pass

    def on_optimization_end(self):
        pass

    def to_dict(self):
        return {"class": "GPy.inference.latent_function_inference.exact_studentt_inference.ExactStudentTInference"}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_state"] = None
        return d

    def inference(self, kern, X, Y, nu, mean_function=None, K=None):
        from scipy.special import digamma
        if K is not None or not isinstance(kern, (Stationary, CombinationKernel)):
            raise NotImplementedError("the MI355X Student-t path evaluates gpy_amd kernels on the device")
        X = np.asarray(X)
        Y = np.asarray(Y, dtype=np.float64)
        m = 0 if mean_function is None else mean_function.f(X)
        R = _lib.f64(Y - m)
        is_sum = isinstance(kern, CombinationKernel)
        Xdev = kern._slice_X(X)
        specs = kern.part_specs() if is_sum else [(kern.kind, kern.ARD, kern._theta(), None)]
        if self._state is None:
            self._state = _DeviceState(self.device)
        st = self._state
        st.ensure_data(Xdev, R)
        st.call_token += 1
        nu = float(np.asarray(nu).ravel()[0])
        extra, tries = 0.0, 0
        while True:                                   # jitchol's ladder (util/linalg.py:56-75)
            info, r = st.ctx.exact_studentt_sum(specs, nu, jitter=1e-8, extra_jitter=extra)
            if info == 0:
                break
            if tries >= self.maxtries:
                raise LinAlgError("not positive definite, even with jitter.")
            extra = (kern.diag_variance() if is_sum else float(specs[0][2][0])) * 1e-6 * 10 ** tries
            tries += 1
        N, beta = Y.shape[0], r["beta"]
        dL_dnu = -N / (nu - 2.0) + digamma(0.5 * (nu + N)) - digamma(0.5 * nu)
        dL_dnu -= np.log(1 + beta / (nu - 2.0))
        dL_dnu += ((nu + N) * beta) / ((nu - 2) * (beta + nu - 2))
        dL_dnu *= 0.5
        sig = kernel_signature(kern)
        dL_dK = DeviceResult(st, _lib.FETCH_DLDK, N, st.call_token, kernel_sig=sig, fused_dtheta=r["dtheta"])
        # beta of the reference's StudentTPosterior is sum(alpha * K alpha) = sum(alpha * R) - jitter |alpha|^2 (posterior.py:345)
        beta_post = beta - (1e-8 + extra) * float(np.sum(r["alpha"] ** 2))
        post = StudentTPosterior(deg_free=nu, beta=beta_post,
                                 woodbury_chol=DeviceResult(st, _lib.FETCH_L, N, st.call_token, fortran_order=True),
                                 woodbury_vector=r["alpha"], K=DeviceResult(st, _lib.FETCH_K, N, st.call_token),
                                 woodbury_inv=DeviceResult(st, _lib.FETCH_KINV, N, st.call_token), state=st)
        return post, r["lml"], {"dL_dK": dL_dK, "dL_dnu": dL_dnu, "dL_dm": r["scale"] * r["alpha"]}

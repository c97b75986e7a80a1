"""Posterior container returned by `ExactGaussianInference.inference` -- mirrors the attributes of the
reference's `PosteriorExact` that `GP` and user code read (`GPy/inference/latent_function_inference/posterior.py:21-77,
131-196,273-302`): `woodbury_chol`, `woodbury_vector`, `woodbury_inv`, `K`, `mean`, `_raw_predict`.
The N x N members are `DeviceResult` proxies (fetched lazily); prediction runs on the device."""
import numpy as np


class PosteriorExact(object):
    def __init__(self, woodbury_chol, woodbury_vector, K, woodbury_inv=None, state=None, prior_mean=0):
        self._woodbury_chol = woodbury_chol
        self._woodbury_vector = woodbury_vector
        self._woodbury_inv = woodbury_inv
        self._K = K
        self._state = state            # device state that produced this posterior (for on-device prediction)
        self._prior_mean = prior_mean
        self._mean = None

    def __getstate__(self):          # device handles never travel; the lazy members materialise themselves
        d = dict(self.__dict__)
        d["_state"] = None
        return d

    @property
    def woodbury_chol(self):
        return self._woodbury_chol

    @property
    def woodbury_vector(self):
        return self._woodbury_vector

    @property
    def woodbury_inv(self):
        return self._woodbury_inv

    @property
    def K(self):
        return self._K

    @property
    def mean(self):
        """K alpha (reference `posterior.py:79-91`)"""
        if self._mean is None:
            self._mean = np.dot(np.asarray(self._K), self._woodbury_vector)
        return self._mean

    def _raw_predict(self, kern, Xnew, pred_var, full_cov=False):
        """mu = K(X*,X) alpha, var = K** - |L^-1 K(X,X*)|^2 (reference `posterior.py:273-302`), on the device."""
        if self._state is None:
            raise RuntimeError("this posterior is not attached to a device context")
        return self._state.predict(kern, Xnew, full_cov=full_cov)

    def predictive_gradients(self, kern, Xnew, pred_var=None):
        """d mean / d Xnew (M x D x Dy) and d var / d Xnew (M x D) (reference `core/gp.py:418-474`): on the device for sums
        of stationary / White / Bias parts, composed on the host from `kern.gradients_X` and the fetched `woodbury_inv`
        for anything else."""
        if self._state is not None:
            try:
                return self._state.predictive_gradients(kern, Xnew)
            except NotImplementedError:
                pass
        X = self._state.X if self._state is not None else None
        if X is None:
            raise RuntimeError("this posterior is not attached to a device context")
        Xnew = np.asarray(Xnew, dtype=np.float64)
        alpha = np.asarray(self.woodbury_vector)
        mean_jac = np.empty((Xnew.shape[0], Xnew.shape[1], alpha.shape[1]))
        for i in range(alpha.shape[1]):
            mean_jac[:, :, i] = kern.gradients_X(alpha[:, i:i + 1].T * np.ones((Xnew.shape[0], 1)), Xnew, X)
        var_jac = kern.gradients_X_diag(np.ones(Xnew.shape[0]), Xnew)
        a2 = -2.0 * np.dot(kern.K(Xnew, X), np.asarray(self.woodbury_inv))
        return mean_jac, var_jac + kern.gradients_X(a2, Xnew, X)

    def covariance_between_points(self, kern, X, X1, X2):
        """K(X1,X2) - (L^-1 K(X,X1))^T (L^-1 K(X,X2)) (reference `posterior.py:109-130`), on the device."""
        if self._state is None:
            raise RuntimeError("this posterior is not attached to a device context")
        return self._state.covariance_between_points(kern, X1, X2)


class StudentTPosterior(PosteriorExact):
    """Posterior of a Student-t PROCESS (reference `posterior.py:338-349`): the Gaussian predictive (co)variance scaled by
    (nu + beta - 2) / (nu + N - 2), beta = sum(alpha * R)."""

    def __init__(self, deg_free, beta=None, **kwargs):
        super(StudentTPosterior, self).__init__(**kwargs)
        self.nu = deg_free
        self._beta = beta            # sum(woodbury_vector * mean) as reduced on the device; None -> recompute on the host

    def _raw_predict(self, kern, Xnew, pred_var, full_cov=False):
        mu, var = super(StudentTPosterior, self)._raw_predict(kern, Xnew, pred_var, full_cov)
        beta = self._beta if self._beta is not None else float(np.sum(self.woodbury_vector * self.mean))
        N = self.woodbury_vector.shape[0]
        return mu, (self.nu + beta - 2.0) / (self.nu + N - 2.0) * var

    def predictive_gradients(self, kern, Xnew, pred_var=None):
        mean_jac, var_jac = super(StudentTPosterior, self).predictive_gradients(kern, Xnew, pred_var)
        beta = self._beta if self._beta is not None else float(np.sum(self.woodbury_vector * self.mean))
        N = self.woodbury_vector.shape[0]
        return mean_jac, (self.nu + beta - 2.0) / (self.nu + N - 2.0) * var_jac

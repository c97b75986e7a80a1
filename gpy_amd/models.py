"""Thin model driver around the hot path: what `GPy.core.GP` / `GPy.models.GPRegression` do with the three
calls this backend replaces (reference `GPy/core/gp.py:38-108,269-365`, `GPy/models/gp_regression.py:29-36`,
`GPy/core/model.py:97-128`).  GPy's own `GP` class can drive the same kernel / inference objects once paramz is
installed; this driver exists so the path is usable and testable without paramz.

Parameter order of the flat vector follows GPy's link order: [kern.variance, kern.lengthscale..., noise variance].
Optimisation runs over log-parameters with SciPy's L-BFGS-B (paramz uses a Logexp transform + the same optimiser
by default; the transform lives outside the boundary either way).
"""
import numpy as np

from .inference import ExactGaussianInference
from .kern import RBF
from .lazy import freeze
from .likelihoods import Gaussian
from .param import Parameterized


class Standardize(object):
    """Output normaliser (reference `GPy/util/normalizer.py:85-112`): zero mean, unit standard deviation per column."""

    def __init__(self):
        self.mean = self.std = None

    def scale_by(self, Y):
        self.mean, self.std = np.nanmean(Y, 0), np.nanstd(Y, 0)
        self.std = np.where(self.std == 0, 1.0, self.std)

    def scaled(self):
        return self.mean is not None

    def normalize(self, Y):
        return (Y - self.mean) / self.std

    def inverse_mean(self, X):
        return X * self.std + self.mean

    def inverse_variance(self, var):
        return var * self.std ** 2

    def inverse_covariance(self, cov):
        return cov[..., np.newaxis] * self.std ** 2

    def to_dict(self):
        return {"class": "GPy.util.normalizer.Standardize", "mean": self.mean.tolist(), "std": self.std.tolist()}


class PredictionCallers(object):
    """The prediction-side callers of the hot path in the reference's `GP` class (`core/gp.py:367-474,601-652,700-790`), shared
    by the exact and the sparse model driver: everything here is a thin wrapper around `_raw_predict` / the posterior, exactly as
    in the reference (`SparseGP` inherits them from `GP`; `_predictive_variable` is X for the exact model and Z for the sparse
    one, `core/gp.py:201-203`, `core/sparse_gp.py:72-74`)."""
    normalizer = None
    output_dim = 1

    def checkgrad(self, verbose=False, step=1e-6, tolerance=1e-3):
        """The gradient check the reference's tests lean on (`paramz.Model.checkgrad`, used e.g. at
        `testing/test_model.py:790-898`): central difference of the objective along a random direction of the flat parameter
        vector against the analytic gradient; True when the ratio is within `tolerance` of one (or both are ~0)."""
        x = self.param_array.copy()
        try:
            g = self.objective_function_gradients().copy()
            dx = np.where(np.random.uniform(size=x.shape) > 0.5, 1.0, -1.0) * step * np.maximum(np.abs(x), 1e-3)
            self.param_array = x + dx
            f1 = self.objective_function()
            self.param_array = x - dx
            f2 = self.objective_function()
        finally:
            self.param_array = x
        num, ana = (f1 - f2) / 2.0, float(np.dot(dx, g))
        if verbose:
            print("checkgrad: numerical %.10e analytic %.10e ratio %.8f" % (num, ana, num / ana if ana != 0 else np.nan))
        if abs(ana) < 1e-14 and abs(num) < 1e-14:
            return True
        return bool(abs(1.0 - num / ana) < tolerance) if ana != 0 else False

    def input_sensitivity(self, summarize=True):
        """(reference `core/gp.py:654-658`)"""
        return self.kern.input_sensitivity(summarize=summarize)

    def predict_noiseless(self, Xnew, full_cov=False):
        """(reference `core/gp.py:367-393`)"""
        return self.predict(Xnew, full_cov=full_cov, include_likelihood=False)

    def predict_quantiles(self, X, quantiles=(2.5, 97.5), Y_metadata=None, kern=None, likelihood=None):
        """(reference `core/gp.py:395-416`)"""
        m, v = self._raw_predict(X, full_cov=False) if kern is None else self._raw_predict(X, full_cov=False, kern=kern)
        likelihood = self.likelihood if likelihood is None else likelihood
        qs = likelihood.predictive_quantiles(m, v, quantiles, Y_metadata=Y_metadata)
        if self.normalizer is not None:
            qs = [self.normalizer.inverse_mean(q) for q in qs]
        return qs

    def log_predictive_density(self, x_test, y_test, Y_metadata=None):
        """(reference `core/gp.py:700-714`)"""
        mu_star, var_star = self._raw_predict(x_test)
        return self.likelihood.log_predictive_density(y_test, mu_star, var_star, Y_metadata=Y_metadata)

    def predictive_gradients(self, Xnew, kern=None):
        """d mean / d X* (N* x Q x D) and d var / d X* (N* x Q) of the latent prediction (reference `core/gp.py:418-474`),
        reduced on the device (`mi355gp_predictive_gradients_sum`)."""
        mean_jac, var_jac = self.posterior.predictive_gradients(self.kern if kern is None else kern, np.asarray(Xnew),
                                                                pred_var=self._predictive_variable)
        if self.normalizer is not None:              # (reference `core/gp.py:467-472`)
            mean_jac = self.normalizer.inverse_mean(mean_jac) - self.normalizer.inverse_mean(0.)
            var_jac = (self.normalizer.inverse_covariance(var_jac) if self.output_dim > 1
                       else self.normalizer.inverse_variance(var_jac))
        return mean_jac, var_jac

    def posterior_samples_f(self, X, size=10, **predict_kwargs):
        """samples of the latent function at X: N* x D x size (reference `core/gp.py:601-629`)"""
        predict_kwargs["full_cov"] = True
        m, v = self._raw_predict(X, **predict_kwargs)
        if self.normalizer is not None:
            m, v = self.normalizer.inverse_mean(m), self.normalizer.inverse_variance(v)

        def sim_one_dim(mm, vv):
            return np.random.multivariate_normal(mm, vv, size).T
        if self.output_dim == 1:
            return sim_one_dim(m.flatten(), v)[:, np.newaxis, :]
        fsim = np.empty((np.asarray(X).shape[0], self.output_dim, size))
        for d in range(self.output_dim):
            fsim[:, d, :] = sim_one_dim(m[:, d], v[:, :, d] if v.ndim == 3 else v)
        return fsim

    def posterior_samples(self, X, size=10, Y_metadata=None, likelihood=None, **predict_kwargs):
        """samples of observations at X (reference `core/gp.py:631-652`)"""
        fsim = self.posterior_samples_f(X, size, **predict_kwargs)
        likelihood = self.likelihood if likelihood is None else likelihood
        for d in range(fsim.shape[1]):
            fsim[:, d] = likelihood.samples(fsim[:, d], Y_metadata=Y_metadata)
        return fsim

    def posterior_covariance_between_points(self, X1, X2, Y_metadata=None, likelihood=None, include_likelihood=True):
        """(reference `core/gp.py:735-790`)"""
        cov = self.posterior.covariance_between_points(self.kern, self._predictive_variable, np.asarray(X1), np.asarray(X2))
        if include_likelihood:
            mean, _ = self._raw_predict(X1, full_cov=True)
            likelihood = self.likelihood if likelihood is None else likelihood
            _, cov = likelihood.predictive_values(mean, cov, full_cov=True, Y_metadata=Y_metadata)
        if self.normalizer is not None:
            cov = self.normalizer.inverse_covariance(cov) if self.output_dim > 1 else self.normalizer.inverse_variance(cov)
        return cov


class GP(PredictionCallers, Parameterized):
    def __init__(self, X, Y, kernel, likelihood, mean_function=None, inference_method=None, name="gp",
                 Y_metadata=None, device=0, normalizer=False):
        super(GP, self).__init__(name)
        # private read-only copies, as GPy wraps X / Y in `ObsAr` (reference `core/gp.py:44-60`): a frozen array is what
        # lets the per-iteration "same data?" check of the inference method be O(1) without ever missing an edit.  New
        # data go through set_XY / set_X / set_Y; `m.X[i] = ...` raises instead of silently leaving the device stale.
        X, Y = freeze(X), freeze(Y)
        assert X.ndim == 2 and Y.ndim == 2
        self.X, self.Y = X, Y
        # (reference `core/gp.py:49-60`): normalizer=True -> Standardize
        self.normalizer = Standardize() if normalizer is True else (None if normalizer is False else normalizer)
        if self.normalizer is not None:
            self.normalizer.scale_by(Y)
            self.Y_normalized = freeze(self.normalizer.normalize(Y))
        else:
            self.Y_normalized = Y
        self.num_data, self.input_dim = X.shape
        self.output_dim = Y.shape[1]
        self.Y_metadata = Y_metadata
        self.kern = kernel
        self.likelihood = likelihood
        self.mean_function = mean_function
        self.inference_method = inference_method or ExactGaussianInference(device=device)
        self.link_parameter(self.kern)
        self.link_parameter(self.likelihood)
        self.posterior = None
        self._log_marginal_likelihood = None
        self.grad_dict = None
        self.parameters_changed()

    def parameters_changed(self):
        """The hot loop body (reference `core/gp.py:278-282`)."""
        self.posterior, self._log_marginal_likelihood, self.grad_dict = self.inference_method.inference(
            self.kern, self.X, self.likelihood, self.Y_normalized, self.mean_function, self.Y_metadata)
        self.likelihood.update_gradients(self.grad_dict["dL_dthetaL"])
        self.kern.update_gradients_full(self.grad_dict["dL_dK"], self.X)
        if self.mean_function is not None:
            self.mean_function.update_gradients(self.grad_dict["dL_dm"], self.X)

    def log_likelihood(self):
        return self._log_marginal_likelihood

    def objective_function(self):
        return -float(self._log_marginal_likelihood)

    def objective_function_gradients(self):
        return -self.gradient

    def set_XY(self, X=None, Y=None):
        """(reference `core/gp.py:188-246`)"""
        if X is not None:
            self.X = freeze(X)                      # a copy: later in-place edits of the caller's buffer do not alias it
            self.num_data = self.X.shape[0]
        if Y is not None:
            self.Y = freeze(Y)
            if self.normalizer is not None:
                self.normalizer.scale_by(self.Y)
                self.Y_normalized = freeze(self.normalizer.normalize(self.Y))
            else:
                self.Y_normalized = self.Y
        self.parameters_changed()

    def _raw_predict(self, Xnew, full_cov=False, kern=None):
        """(reference `core/gp.py:290-306`)"""
        mu, var = self.posterior._raw_predict(kern=self.kern if kern is None else kern, Xnew=np.asarray(Xnew),
                                              pred_var=self.X, full_cov=full_cov)
        if self.mean_function is not None:
            mu = mu + self.mean_function.f(Xnew)
        return mu, var

    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """(reference `core/gp.py:308-365`, same argument order)"""
        mu, var = self._raw_predict(Xnew, full_cov=full_cov, kern=kern)
        if include_likelihood:
            likelihood = self.likelihood if likelihood is None else likelihood
            mu, var = likelihood.predictive_values(mu, var, full_cov=full_cov,
                                                   Y_metadata=self.Y_metadata if Y_metadata is None else Y_metadata)
        if self.normalizer is not None:              # (reference `core/gp.py:353-363`)
            mu = self.normalizer.inverse_mean(mu)
            if full_cov and mu.shape[1] > 1:
                var = self.normalizer.inverse_covariance(var)
            else:
                var = self.normalizer.inverse_variance(var)
        return mu, var

    @property
    def _predictive_variable(self):
        return self.X

    def set_X(self, X):
        """(reference `core/gp.py:251-258`)"""
        self.set_XY(X=X)

    def set_Y(self, Y):
        """(reference `core/gp.py:260-267`)"""
        self.set_XY(Y=Y)

    def optimize(self, max_iters=1000, messages=False, gtol=1e-6):
        """L-BFGS-B on the negative log marginal likelihood in log-parameter space."""
        from scipy.optimize import minimize
        x0 = np.log(self.param_array)

        def f(z):
            try:
                self.param_array = np.exp(z)
            except np.linalg.LinAlgError:
                return 1e300, np.zeros_like(z)
            return self.objective_function(), self.objective_function_gradients() * np.exp(z)
        res = minimize(f, x0, jac=True, method="L-BFGS-B", options={"maxiter": max_iters, "gtol": gtol,
                                                                     "disp": bool(messages)})
        self.param_array = np.exp(res.x)
        return res


class GPRegression(GP):
    """Gaussian-process regression with Gaussian noise (reference `GPy/models/gp_regression.py:29-36`)."""

    def __init__(self, X, Y, kernel=None, Y_metadata=None, normalizer=None, noise_var=1., mean_function=None,
                 device=0):
        if kernel is None:
            kernel = RBF(np.asarray(X).shape[1], device=device)
        super(GPRegression, self).__init__(X, Y, kernel, Gaussian(variance=noise_var), name="GP regression",
                                           Y_metadata=Y_metadata, mean_function=mean_function, device=device,
                                           normalizer=bool(normalizer) if normalizer in (None, True, False)
                                           else normalizer)


class GPHeteroscedasticRegression(GP):
    """One Gaussian noise variance per data point (reference `GPy/models/gp_heteroscedastic_regression.py:10-40`): the
    length-N noise vector goes to the device with the inference call and the N noise gradients come back as diag(dL_dK)."""

    def __init__(self, X, Y, kernel=None, Y_metadata=None, device=0):
        from .likelihoods import HeteroscedasticGaussian
        Ny = np.asarray(Y).shape[0]
        if Y_metadata is None:
            Y_metadata = {"output_index": np.arange(Ny)[:, None]}
        else:
            assert np.asarray(Y_metadata["output_index"]).shape[0] == Ny
        if kernel is None:
            kernel = RBF(np.asarray(X).shape[1], device=device)
        super(GPHeteroscedasticRegression, self).__init__(X, Y, kernel, HeteroscedasticGaussian(Y_metadata),
                                                          name="gp_heteroscedastic_regression", Y_metadata=Y_metadata,
                                                          device=device)

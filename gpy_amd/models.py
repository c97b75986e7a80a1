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
from .likelihoods import Gaussian
from .param import Parameterized


class GP(Parameterized):
    def __init__(self, X, Y, kernel, likelihood, mean_function=None, inference_method=None, name="gp",
                 Y_metadata=None, device=0):
        super(GP, self).__init__(name)
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        assert X.ndim == 2 and Y.ndim == 2
        self.X, self.Y = X, Y
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
            self.X = np.asarray(X, dtype=np.float64)
            self.num_data = self.X.shape[0]
        if Y is not None:
            self.Y = self.Y_normalized = np.asarray(Y, dtype=np.float64)
        self.parameters_changed()

    def _raw_predict(self, Xnew, full_cov=False, kern=None):
        mu, var = self.posterior._raw_predict(kern=self.kern if kern is None else kern, Xnew=np.asarray(Xnew),
                                              pred_var=self.X, full_cov=full_cov)
        if self.mean_function is not None:
            mu = mu + self.mean_function.f(Xnew)
        return mu, var

    def predict(self, Xnew, full_cov=False, include_likelihood=True):
        """(reference `core/gp.py:308-365`)"""
        mu, var = self._raw_predict(Xnew, full_cov=full_cov)
        if include_likelihood:
            mu, var = self.likelihood.predictive_values(mu, var, full_cov=full_cov, Y_metadata=self.Y_metadata)
        return mu, var

    def predict_noiseless(self, Xnew, full_cov=False):
        return self.predict(Xnew, full_cov=full_cov, include_likelihood=False)

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

    def __init__(self, X, Y, kernel=None, Y_metadata=None, noise_var=1., mean_function=None, device=0):
        if kernel is None:
            kernel = RBF(np.asarray(X).shape[1], device=device)
        super(GPRegression, self).__init__(X, Y, kernel, Gaussian(variance=noise_var), name="GP regression",
                                           Y_metadata=Y_metadata, mean_function=mean_function, device=device)

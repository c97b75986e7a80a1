"""Gaussian likelihood: the caller-side contract of the exact-inference hot path
(reference `GPy/likelihoods/gaussian.py:43,69-79,102-110`).  O(1)/O(N) host bookkeeping only."""
import numpy as np

from .param import Param, Parameterized


class Gaussian(Parameterized):
    def __init__(self, variance=1., name="Gaussian_noise"):
        super(Gaussian, self).__init__(name)
        self.variance = Param("variance", variance)
        self.link_parameter(self.variance)

    def gaussian_variance(self, Y_metadata=None):
        return self.variance

    def exact_inference_gradients(self, dL_dKdiag, Y_metadata=None):
        return np.sum(dL_dKdiag)

    def update_gradients(self, grad):
        self.variance.gradient = grad

    def predictive_values(self, mu, var, full_cov=False, Y_metadata=None):
        if full_cov:
            var = var + np.eye(var.shape[0]) * float(self.variance.values[0])
        else:
            var = var + float(self.variance.values[0])
        return mu, var

    def predictive_quantiles(self, mu, var, quantiles, Y_metadata=None):
        """(reference `gaussian.py:118-119`)"""
        from scipy import stats
        return [stats.norm.ppf(q / 100.) * np.sqrt(var + float(self.variance.values[0])) + mu for q in quantiles]

    def log_predictive_density(self, y_test, mu_star, var_star, Y_metadata=None):
        """independent Gaussian predictive densities (reference `gaussian.py:329-334`)"""
        v = var_star + float(self.variance.values[0])
        return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(v) - 0.5 * np.square(y_test - mu_star) / v

    def samples(self, gp, Y_metadata=None):
        """observations drawn around latent values (reference `gaussian.py:316-327`)"""
        gp = np.asarray(gp)
        return gp + np.sqrt(float(self.variance.values[0])) * np.random.normal(size=gp.shape)

    def to_dict(self):
        return {"class": "GPy.likelihoods.Gaussian", "name": self.name, "variance": self.variance.values.tolist()}


class HeteroscedasticGaussian(Gaussian):
    """One noise variance per data point (reference `GPy/likelihoods/gaussian.py:347-371`): `variance` has as many entries
    as `Y_metadata['output_index']`, `gaussian_variance` hands the length-N vector to the inference call (the device adds it
    to the diagonal of K) and the noise gradients are the matching entries of diag(dL_dK), which the device returns as an
    N-vector (`mi355gp_exact_inference(..., diag_dLdK_out)`)."""

    def __init__(self, Y_metadata, variance=1., name="het_Gauss"):
        n = np.asarray(Y_metadata["output_index"]).shape
        Parameterized.__init__(self, name)
        self.variance = Param("variance", np.ones(n).ravel() * variance)
        self.link_parameter(self.variance)

    def gaussian_variance(self, Y_metadata=None):
        return self.variance.values[np.asarray(Y_metadata["output_index"]).flatten()]

    def exact_inference_gradients(self, dL_dKdiag, Y_metadata=None):
        return np.asarray(dL_dKdiag).reshape(-1)[np.asarray(Y_metadata["output_index"]).flatten()]

    def predictive_values(self, mu, var, full_cov=False, Y_metadata=None):
        s = self.variance.values[np.asarray(Y_metadata["output_index"]).flatten()]
        if full_cov:
            return mu, var + np.eye(var.shape[0]) * s
        return mu, var + s[:, None]

    def predictive_quantiles(self, mu, var, quantiles, Y_metadata=None):
        """(reference `gaussian.py:375-377`)"""
        from scipy import stats
        s = self.variance.values[np.asarray(Y_metadata["output_index"]).flatten()]
        return [stats.norm.ppf(q / 100.) * np.sqrt(var + s[:, None]) + mu for q in quantiles]

    def to_dict(self):
        return {"class": "GPy.likelihoods.HeteroscedasticGaussian", "name": self.name,
                "variance": self.variance.values.tolist()}

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

    def to_dict(self):
        return {"class": "GPy.likelihoods.Gaussian", "name": self.name, "variance": self.variance.values.tolist()}

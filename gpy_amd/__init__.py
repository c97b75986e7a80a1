"""gpy_amd -- MI355X-native (gfx950) backend for GPy's exact-GP hot path.

    from gpy_amd import RBF, Matern52, Gaussian, ExactGaussianInference, GPRegression

Host code is Python + ctypes over the C-ABI in include/mi355gp.h; every array operation of the path
(K build, Cholesky, triangular inverse, Ky^-1, alpha, gradient reductions, prediction) is a hand-written
HIP kernel in gpy_amd/csrc.  No PyTorch, no NumPy fallback: without an MI355X the compute calls raise.
"""
from . import _lib
from ._lib import MI355GPError, build, device_count
from .inference import ExactGaussianInference, ExactStudentTInference
from .kern import RBF, Add, Prod, Bias, ExpQuad, Exponential, Matern32, Matern52, Stationary, White
from .likelihoods import Gaussian, HeteroscedasticGaussian
from .models import GP, GPRegression
from .posterior import PosteriorExact, StudentTPosterior
from .sparse import SparseGP, SparseGPRegression, VarDTC

__all__ = ["RBF", "ExpQuad", "HeteroscedasticGaussian", "StudentTPosterior", "Matern52", "Matern32", "Exponential", "Stationary", "White", "Bias", "Add", "Prod", "Gaussian", "ExactGaussianInference", "ExactStudentTInference",
           "PosteriorExact", "GP", "GPRegression", "VarDTC", "SparseGP", "SparseGPRegression", "MI355GPError", "build", "device_count"]

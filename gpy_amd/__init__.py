"""gpy_amd -- MI355X-native (gfx950) backend for GPy's exact-GP hot path.

    from gpy_amd import RBF, Matern52, Gaussian, ExactGaussianInference, GPRegression

Host code is Python + ctypes over the C-ABI in include/mi355gp.h; every array operation of the path
(K build, Cholesky, triangular inverse, Ky^-1, alpha, gradient reductions, prediction) is a hand-written
HIP kernel in gpy_amd/csrc.  No PyTorch, no NumPy fallback: without an MI355X the compute calls raise.
"""
from . import _lib
from ._lib import MI355GPError, build, device_count
from .inference import ExactGaussianInference, ExactStudentTInference
from .kern import OU, RBF, Add, Prod, Bias, ExpQuad, Exponential, Matern32, Matern52, Stationary, White
from .likelihoods import Gaussian, HeteroscedasticGaussian
from .models import GP, GPHeteroscedasticRegression, GPRegression
from .posterior import PosteriorExact, StudentTPosterior
from .sparse import SparseGP, SparseGPRegression, VarDTC

__all__ = ["RBF", "OU", "ExpQuad", "HeteroscedasticGaussian", "StudentTPosterior", "Matern52", "Matern32", "Exponential", "Stationary", "White", "Bias", "Add", "Prod", "Gaussian", "ExactGaussianInference", "ExactStudentTInference",
           "PosteriorExact", "GP", "GPRegression", "GPHeteroscedasticRegression", "VarDTC", "SparseGP", "SparseGPRegression", "MI355GPError", "build", "device_count"]

# GPy's import paths, so that `import gpy_amd as GPy` reads like the reference on this path:
#   GPy.kern.RBF, GPy.likelihoods.Gaussian, GPy.models.GPRegression / SparseGPRegression / GPHeteroscedasticRegression,
#   GPy.core.GP / SparseGP, GPy.inference.latent_function_inference.ExactGaussianInference / VarDTC
from . import inference, kern, likelihoods, linalg, models, sparse  # noqa: E402
import types as _types  # noqa: E402

models.SparseGPRegression = SparseGPRegression
core = _types.SimpleNamespace(GP=GP, SparseGP=SparseGP)
util = _types.SimpleNamespace(linalg=linalg)
inference.latent_function_inference = _types.SimpleNamespace(
    ExactGaussianInference=ExactGaussianInference, ExactStudentTInference=ExactStudentTInference, VarDTC=VarDTC,
    PosteriorExact=PosteriorExact, StudentTPosterior=StudentTPosterior,
    exact_gaussian_inference=_types.SimpleNamespace(ExactGaussianInference=ExactGaussianInference),
    var_dtc=_types.SimpleNamespace(VarDTC=VarDTC))

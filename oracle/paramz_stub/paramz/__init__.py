"""TEST-ONLY stand-in for the third-party `paramz` package (NOT product code).

GPy (`/root/reference`) depends on `paramz>0.9.6` (reference `setup.py:146`), which is neither
vendored nor installable here (no network).  This ~100-line stand-in provides only the names the
reference's hot-path modules touch at import/run time, so that the reference's UNMODIFIED files
(`GPy/kern/src/stationary.py`, `rbf.py`, `inference/.../exact_gaussian_inference.py`,
`likelihoods/gaussian.py`, `util/linalg.py` ...) can be executed by `oracle/ref_loader.py` to
validate the oracle and to generate `tests/golden/*.npz`.  Observers, transformations, optimizers,
caching and pretty printing are deliberately absent.
"""
import numpy as np
from .core.parameter_core import Parameterizable


class ObsAr(np.ndarray):
    def __new__(cls, input_array, *a, **kw):
        return np.atleast_1d(np.asarray(input_array, dtype=float)).view(cls)

    @property
    def values(self):
        return self.view(np.ndarray)

    def copy(self):
        return np.ndarray.copy(self.view(np.ndarray)).view(type(self))


class Param(ObsAr, Parameterizable):
    def __new__(cls, name, input_array, *a, **kw):
        obj = np.atleast_1d(np.array(input_array, dtype=float)).view(cls)
        obj.name = name
        obj._grad = np.zeros(obj.shape)
        return obj

    def __init__(self, name, input_array, *a, **kw):
        self.parameters = []

    def __array_finalize__(self, obj):
        self.name = getattr(obj, 'name', None)
        self._grad = getattr(obj, '_grad', None)

    @property
    def gradient(self):
        return self._grad

    @gradient.setter
    def gradient(self, g):
        self._grad = np.broadcast_to(np.asarray(g, dtype=float), self.shape).copy()

    def constrain_positive(self, *a, **k): pass
    def constrain_fixed(self, *a, **k): pass
    fix = constrain_fixed
    def constrain_bounded(self, *a, **k): pass
    def constrain(self, *a, **k): pass


class Parameterized(Parameterizable):
    pass


class Model(Parameterized):
    pass


def load(*a, **k):
    raise NotImplementedError("test-only paramz stand-in")

"""Minimal parameter containers used when GPy/paramz are not importable.

GPy's kernels expose `Param` arrays (`GPy/kern/src/stationary.py:78-81`) whose `.gradient` the hot path
writes (`stationary.py:199,208-213`).  This module provides just that contract: an ndarray subclass with a
same-shaped `.gradient`, plus a tiny `Parameterized` that can flatten / restore its parameters.  It is
host-side bookkeeping (O(#params)); all array math of the hot path runs in libmi355gp.so.
"""
import numpy as np


class Param(np.ndarray):
    def __new__(cls, name, value, positive=True):
        obj = np.atleast_1d(np.array(value, dtype=np.float64)).view(cls)
        obj.name = name
        obj.positive = positive
        obj._gradient = np.zeros(obj.shape)
        return obj

    def __array_finalize__(self, obj):
        self.name = getattr(obj, "name", None)
        self.positive = getattr(obj, "positive", True)
        self._gradient = getattr(obj, "_gradient", None)

    def __reduce__(self):
        base = super(Param, self).__reduce__()
        return (base[0], base[1], base[2] + (self.name, self.positive, self._gradient))

    def __setstate__(self, state):
        self.name, self.positive, self._gradient = state[-3:]
        super(Param, self).__setstate__(state[:-3])

    @property
    def values(self):
        return self.view(np.ndarray)

    @property
    def gradient(self):
        return self._gradient

    @gradient.setter
    def gradient(self, g):
        self._gradient = np.broadcast_to(np.asarray(g, dtype=np.float64), self.shape).copy()


class Parameterized(object):
    """Holds an ordered list of Params / child Parameterized objects (GPy link order)."""

    def __init__(self, name=None):
        self.name = name
        self.parameters = []

    def link_parameter(self, p, index=None):
        if index is None:
            self.parameters.append(p)
        else:
            self.parameters.insert(index, p)

    def link_parameters(self, *ps):
        for p in ps:
            self.link_parameter(p)

    def unlink_parameter(self, p):
        self.parameters = [q for q in self.parameters if q is not p]

    def flattened_parameters(self):
        out = []
        for p in self.parameters:
            if isinstance(p, Parameterized):
                out.extend(p.flattened_parameters())
            else:
                out.append(p)
        return out

    @property
    def size(self):
        return int(sum(p.size for p in self.flattened_parameters()))

    @property
    def param_array(self):
        ps = self.flattened_parameters()
        return np.concatenate([p.values.ravel() for p in ps]) if ps else np.zeros(0)

    @param_array.setter
    def param_array(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        i = 0
        for p in self.flattened_parameters():
            p[...] = x[i:i + p.size].reshape(p.shape)
            i += p.size
        self._propagate_changed()

    def _propagate_changed(self):
        for p in self.parameters:
            if isinstance(p, Parameterized):
                p._propagate_changed()
        self.parameters_changed()

    @property
    def gradient(self):
        ps = self.flattened_parameters()
        return np.concatenate([p.gradient.ravel() for p in ps]) if ps else np.zeros(0)

    def parameter_names(self):
        names = []
        for p in self.parameters:
            if isinstance(p, Parameterized):
                names.extend("%s.%s" % (p.name, n) for n in p.parameter_names())
            else:
                names.extend([p.name] if p.size == 1 else ["%s[%d]" % (p.name, i) for i in range(p.size)])
        return names

    def parameters_changed(self):
        pass

"""Lazy host views of the N x N results that stay resident in HBM.

GPy's contract hands `dL_dK` (N x N) from the inference method to the kernel and stores `woodbury_chol` / `K`
in the Posterior (reference `core/gp.py:278-280`, `exact_gaussian_inference.py:74`).  Shipping those over PCIe
every optimiser iteration would dominate the run, so the drop-in returns `DeviceResult` proxies: they behave
as arrays for foreign consumers (`np.asarray`, indexing, arithmetic -> one fetch, cached) while gpy_amd's own
kernels recognise them and never leave the device.
"""
import numpy as np

from . import _lib


class DeviceResult(object):
    __array_priority__ = 100.0

    def __init__(self, ctx, which, n, token, fortran_order=False, kernel_sig=None, fused_dtheta=None):
        self._ctx = ctx
        self._which = which
        self._n = n
        self._token = token          # inference call counter: a stale proxy must not read newer buffers
        self._fortran = fortran_order
        self._host = None
        self._kernel_sig = kernel_sig
        self.fused_dtheta = fused_dtheta

    shape = property(lambda self: (self._n, self._n))
    ndim = 2
    dtype = np.dtype(np.float64)
    size = property(lambda self: self._n * self._n)

    def __getstate__(self):
        """Pickling materialises the host copy (while the device still holds it) and drops the device handle."""
        d = dict(self.__dict__)
        if d["_host"] is None and d["_ctx"] is not None:
            try:
                d["_host"] = self.fetch()
            except Exception:
                d["_host"] = None
        d["_ctx"] = None
        return d

    def matches_kernel(self, kern):
        return self.fused_dtheta is not None and self._kernel_sig == kernel_signature(kern)

    def fetch(self):
        if self._host is None:
            if self._ctx is None:
                raise RuntimeError("this result was detached from its device context (unpickled) before being fetched")
            if self._ctx.call_token != self._token:
                raise RuntimeError("this device-resident result was overwritten by a later inference call; "
                                   "materialise it (np.asarray) before re-running inference")
            self._host = self._ctx.fetch(self._which, fortran_order=self._fortran)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.fetch()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.fetch()[idx]

    def __len__(self):
        return self._n

    @property
    def T(self):
        return self.fetch().T

    def copy(self):
        return self.fetch().copy()

    def sum(self, *a, **k):
        return self.fetch().sum(*a, **k)

    def dot(self, other):
        return self.fetch().dot(other)

    def __repr__(self):
        return "DeviceResult(which=%d, shape=%r, %s)" % (self._which, self.shape,
                                                         "materialised" if self._host is not None else "on device")


def _binop(name):
    def f(self, other):
        return getattr(self.fetch(), name)(other)
    return f


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__",
           "__matmul__", "__rmatmul__", "__neg__"):
    if _n == "__neg__":
        setattr(DeviceResult, _n, lambda self: -self.fetch())
    else:
        setattr(DeviceResult, _n, _binop(_n))


def kernel_signature(kern):
    """Identity of a kernel evaluation: class, ARD flag, active dims and the exact parameter bits (per part for sums)."""
    parts = getattr(kern, "parts", None)
    if parts is not None:
        return (type(kern).__name__,) + tuple(kernel_signature(p) for p in parts)
    return (kern.kind, bool(kern.ARD), tuple(int(i) for i in kern.active_dims), kern._theta().tobytes())

"""Lazy host views of the N x N results that stay resident in HBM.

GPy's contract hands `dL_dK` (N x N) from the inference method to the kernel and stores `woodbury_chol` / `K`
in the Posterior (reference `core/gp.py:278-280`, `exact_gaussian_inference.py:74`).  Shipping those over PCIe
every optimiser iteration would dominate the run, so the drop-in returns `DeviceResult` proxies: they behave
as arrays for foreign consumers (`np.asarray`, indexing, arithmetic -> one fetch, cached) while gpy_amd's own
kernels recognise them and never leave the device.
"""
import numpy as np

from . import _lib


def frozen(a):
    """True when nobody can write through `a` or any array it is a view of: every ndarray in its base chain is
    read-only and the chain ends in memory NumPy itself owns.  Only such arrays can be recognised by IDENTITY from one
    call to the next; GPy gets the same guarantee from `ObsAr` observers (reference `core/gp.py:44-60`)."""
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None


def freeze(a):
    """A private, read-only float64 C-contiguous copy of `a` (what the model drivers keep as X / Y)."""
    b = np.array(a, dtype=np.float64, order="C", copy=True)
    b.setflags(write=False)
    return b


class ArrayIdentity(object):
    """Remembers one host array so that a later call can tell whether it was handed THE SAME DATA.
    A frozen array (see `frozen`) is remembered by object identity: O(1) per call, and it cannot have been edited in place.
    Anything else is remembered as a private copy and compared element by element (O(size), like paramz would re-run the
    computation for a plain ndarray): a sampled fingerprint can miss an in-place edit, identity of a writable buffer can
    alias a different array after the allocator reuses the address."""

    __slots__ = ("obj", "copy")

    def __init__(self, a):
        if a is None:
            self.obj = self.copy = None
        elif frozen(a):
            self.obj, self.copy = a, None
        else:
            self.obj, self.copy = None, np.array(a, copy=True)

    def matches(self, b):
        if b is None:
            return self.obj is None and self.copy is None
        if self.obj is not None:
            if b is self.obj and frozen(b):
                return True
            ref = self.obj
        elif self.copy is not None:
            ref = self.copy
        else:
            return False
        b = np.asarray(b)
        return ref.shape == b.shape and ref.dtype == b.dtype and np.array_equal(ref, b)

    def value(self):
        return self.obj if self.obj is not None else self.copy


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

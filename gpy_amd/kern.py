"""Stationary covariance functions backed by libmi355gp.so -- drop-in for the hot-path methods of
`GPy.kern.RBF / Matern52 / Matern32 / Exponential`:

    K(X, X2=None), Kdiag(X), update_gradients_full(dL_dK, X, X2=None), update_gradients_diag(dL_dKdiag, X)

Same constructor arguments, parameter names (`variance`, `lengthscale`, `inv_lengthscale`), link order and
`.gradient` side effects as the reference (`GPy/kern/src/stationary.py:60-81,105-115,170-213`,
`GPy/kern/src/rbf.py:22-33,373-375`); `active_dims` slicing follows `GPy/kern/src/kern.py:112-117`.
All array math runs in hand-written HIP kernels (csrc/kern.hip); there is no NumPy fallback.
"""
import numpy as np

from . import _lib
from .lazy import ArrayIdentity, DeviceResult
from .param import Param, Parameterized


class _KCache(object):
    """`@Cache_this(limit=3)` of the reference's `Stationary.K` (`stationary.py:96-105`, paramz `Cacher`): the last
    `limit` results keyed by (X, X2) and the exact parameter bits, so the K that the inference step evaluated is not
    recomputed (upload + build + download) when the gradient / prediction step asks for it again.
    paramz caches only for `Observable` inputs and invalidates through their observers.  Here an input is recognised either
    by identity -- only if it is FROZEN (read-only through its whole base chain, which is how the model drivers hold X) --
    or by a full element-wise comparison with a private copy (O(N D), nothing next to the N^2 D build); never by a sampled
    fingerprint or by the address of a writable buffer.  The cached matrix is shared between callers and therefore
    read-only: `K = k.K(X).copy()` before editing it in place (the reference hands out its cache entry writable, and an
    in-place edit silently corrupts later hits)."""

    def __init__(self, limit=3):
        self.limit = limit
        self.entries = []            # [(idX, idX2, theta bytes, value)], most recent last

    def get(self, X, X2, theta, compute):
        tb = theta.tobytes()
        for i, (ix, ix2, t, v) in enumerate(self.entries):
            if t == tb and ix.matches(X) and ix2.matches(X2):
                self.entries.append(self.entries.pop(i))
                return v
        v = compute()
        v.setflags(write=False)      # shared between callers, like paramz's cached arrays
        self.entries.append((ArrayIdentity(X), ArrayIdentity(X2), tb, v))
        if len(self.entries) > self.limit:
            self.entries.pop(0)
        return v

    def clear(self):
        self.entries = []


class Stationary(Parameterized):
    kind = None                # name understood by the C-ABI
    _gpy_class = None          # "class" string for to_dict (resolvable by GPy's loader)
    _support_GPU = True

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name=None,
                 useGPU=True, device=0):
        super(Stationary, self).__init__(name or self.kind)
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        self.device = device
        self.useGPU = True
        if active_dims is None:
            active_dims = np.arange(self.input_dim)
        self.active_dims = np.atleast_1d(np.asarray(active_dims, dtype=np.int_))
        assert self.active_dims.size == self.input_dim, "input_dim=%d does not match len(active_dims)=%d" % (
            self.input_dim, self.active_dims.size)
        if not self.ARD:
            if lengthscale is None:
                lengthscale = np.ones(1)
            else:
                lengthscale = np.asarray(lengthscale, dtype=float)
                assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        else:
            if lengthscale is not None:
                lengthscale = np.asarray(lengthscale, dtype=float)
                assert lengthscale.size in [1, self.input_dim], "Bad number of lengthscales"
                if lengthscale.size != self.input_dim:
                    lengthscale = np.ones(self.input_dim) * lengthscale
            else:
                lengthscale = np.ones(self.input_dim)
        self.variance = Param("variance", variance)
        self.lengthscale = Param("lengthscale", lengthscale)
        assert self.variance.size == 1
        self.link_parameters(self.variance, self.lengthscale)
        self._K_cache = _KCache(limit=3)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_K_cache"] = _KCache(limit=3)
        return d

    # ---- helpers ---------------------------------------------------------------------------------
    def _slice_X(self, X):
        X = np.asarray(X)
        if X.shape[1] == self.input_dim and np.array_equal(self.active_dims, np.arange(self.input_dim)):
            return _lib.f64(X)
        assert X.shape[1] > self.active_dims.max(), "At least %d dimensional X needed, X.shape=%r" % (
            self.active_dims.max() + 1, X.shape)
        return _lib.f64(X[:, self.active_dims])

    def _theta(self):
        return _lib.theta_vec(self.variance.values, self.lengthscale.values, self.ARD, self.input_dim)

    # ---- the hot-path interface -----------------------------------------------------------------------
    def K(self, X, X2=None):
        """Covariance matrix K(X, X2) (reference `stationary.py:105-115`, `@Cache_this(limit=3)` :105)."""
        X = np.asarray(X)
        X2 = None if X2 is None else np.asarray(X2)
        theta = self._theta()

        def compute():
            Xs = self._slice_X(X)
            X2s = None if X2 is None else self._slice_X(X2)
            return _lib.kern_K(self.kind, self.ARD, theta, Xs, X2s, device=self.device)
        return self._K_cache.get(X, X2, theta, compute)

    def Kdiag(self, X):
        """(reference `stationary.py:170-173`)"""
        return _lib.kern_Kdiag(self.kind, self._theta(), np.asarray(X).shape[0])

    def update_gradients_full(self, dL_dK, X, X2=None):
        """Writes `self.variance.gradient` and `self.lengthscale.gradient` (reference `stationary.py:193-213`).

        When `dL_dK` is the device-resident result of `gpy_amd.ExactGaussianInference` for this kernel the
        gradients were already reduced on the GPU in the same pass and are simply installed."""
        if isinstance(dL_dK, DeviceResult) and X2 is None and dL_dK.matches_kernel(self):
            g = dL_dK.fused_dtheta
        else:
            g = _lib.update_gradients_full(self.kind, self.ARD, self._theta(), np.asarray(dL_dK), self._slice_X(X),
                                           None if X2 is None else self._slice_X(X2), device=self.device)
        self._install_gradients(g)

    def _install_gradients(self, g):
        self.variance.gradient = g[0]
        self.lengthscale.gradient = g[1:] if self.ARD else g[1]

    def update_gradients_diag(self, dL_dKdiag, X):
        """(reference `stationary.py:182-191`)"""
        self.variance.gradient = np.sum(dL_dKdiag)
        self.lengthscale.gradient = 0.

    def update_gradients_direct(self, dL_dVar, dL_dLen):
        """install gradients computed elsewhere (reference `stationary.py:215-223`)"""
        self.variance.gradient = dL_dVar
        self.lengthscale.gradient = dL_dLen

    def input_sensitivity(self, summarize=True):
        """variance / lengthscale^2 per input dimension (reference `stationary.py:363-364`)"""
        return float(self.variance.values[0]) * np.ones(self.input_dim) / np.asarray(self.lengthscale.values) ** 2

    def gradients_X(self, dL_dK, X, X2=None):
        """dL/dX from dL_dK (reference `stationary.py:245-252,330-358`), reduced on the device."""
        g = _lib.gradients_X(self.kind, self.ARD, self._theta(), np.asarray(dL_dK), self._slice_X(X),
                             None if X2 is None else self._slice_X(X2), device=self.device)
        if g.shape[1] == np.asarray(X).shape[1]:
            return g
        full = np.zeros(np.asarray(X).shape)          # active_dims slicing (kernel_slice_operations.py:113-136)
        full[:, self.active_dims] = g
        return full

    def gradients_X_diag(self, dL_dKdiag, X):
        """(reference `stationary.py:360-361`)"""
        return np.zeros(np.asarray(X).shape)

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])

    def reset_gradients(self):
        self.variance.gradient = 0.
        self.lengthscale.gradient = np.zeros(self.input_dim) if self.ARD else 0.

    # ---- bookkeeping ----------------------------------------------------------------------------------
    def to_dict(self):
        """JSON-serialisable description with the reference's "class" string (`stationary.py:83-88`)."""
        return {"class": self._gpy_class, "name": self.name, "input_dim": self.input_dim,
                "active_dims": self.active_dims.tolist(), "variance": self.variance.values.tolist(),
                "lengthscale": self.lengthscale.values.tolist(), "ARD": self.ARD, "useGPU": True}

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        d.pop("class", None)
        d.pop("useGPU", None)
        return cls(**d)

    def copy(self):
        return self.__class__.from_dict(self.to_dict())


class RBF(Stationary):
    """k(r) = variance * exp(-r^2/2)  (reference `GPy/kern/src/rbf.py:51-52`); `inv_l=True` re-parameterises by
    the inverse squared lengthscale exactly like the reference (`rbf.py:29-33,328-330,373-375`)."""
    kind = "rbf"
    _gpy_class = "GPy.kern.RBF"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="rbf",
                 useGPU=True, inv_l=False, device=0):
        super(RBF, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, useGPU, device)
        self.use_invLengthscale = bool(inv_l)
        if self.use_invLengthscale:
            self.unlink_parameter(self.lengthscale)
            self.inv_l = Param("inv_lengthscale", 1. / self.lengthscale.values ** 2)
            self.link_parameter(self.inv_l)

    def parameters_changed(self):
        if self.use_invLengthscale:
            self.lengthscale[:] = 1. / np.sqrt(self.inv_l.values + 1e-200)

    def _install_gradients(self, g):
        super(RBF, self)._install_gradients(g)
        if self.use_invLengthscale:
            self.inv_l.gradient = self.lengthscale.gradient * (self.lengthscale.values ** 3 / -2.)

    def update_gradients_diag(self, dL_dKdiag, X):
        super(RBF, self).update_gradients_diag(dL_dKdiag, X)
        if self.use_invLengthscale:
            self.inv_l.gradient = self.lengthscale.gradient * (self.lengthscale.values ** 3 / -2.)

    def to_dict(self):
        d = super(RBF, self).to_dict()
        d["inv_l"] = self.use_invLengthscale
        return d


class ExpQuad(Stationary):
    """The exponentiated quadratic k(r) = variance * exp(-r^2/2) (reference `stationary.py:623-662`): the same
    function as `RBF` without the psi-statistics / `inv_l` extras; serialises as "GPy.kern.ExpQuad"."""
    kind = "rbf"
    _gpy_class = "GPy.kern.ExpQuad"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="ExpQuad", **kw):
        super(ExpQuad, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, **kw)


class Matern52(Stationary):
    """k(r) = variance (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)  (reference `stationary.py:585-589`)."""
    kind = "matern52"
    _gpy_class = "GPy.kern.Matern52"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="Mat52", **kw):
        super(Matern52, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, **kw)


class Matern32(Stationary):
    """k(r) = variance (1 + sqrt3 r) exp(-sqrt3 r)  (reference `stationary.py:488-492`)."""
    kind = "matern32"
    _gpy_class = "GPy.kern.Matern32"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="Mat32", **kw):
        super(Matern32, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, **kw)


class Exponential(Stationary):
    """k(r) = variance exp(-r)  (reference `stationary.py:382-386`)."""
    kind = "exponential"
    _gpy_class = "GPy.kern.Exponential"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="Exponential", **kw):
        super(Exponential, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, **kw)


class OU(Exponential):
    """Ornstein-Uhlenbeck = the exponential kernel under its other name (reference `stationary.py:416-454`)."""
    _gpy_class = "GPy.kern.OU"

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="OU", **kw):
        super(OU, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, **kw)


class Static(Parameterized):
    """White / Bias (reference `GPy/kern/src/static.py:10-60`): one `variance` parameter, no input dependence."""
    kind = None
    _gpy_class = None

    def __init__(self, input_dim, variance=1., active_dims=None, name=None, device=0):
        super(Static, self).__init__(name or self.kind)
        self.input_dim = int(input_dim)
        self.device = device
        self.ARD = False
        self.active_dims = np.arange(self.input_dim) if active_dims is None else np.atleast_1d(
            np.asarray(active_dims, dtype=np.int_))
        self.variance = Param("variance", variance)
        self.link_parameter(self.variance)

    def _theta(self):
        return np.array([float(self.variance.values[0])])

    def Kdiag(self, X):
        return np.full(np.asarray(X).shape[0], float(self.variance.values[0]))

    def update_gradients_diag(self, dL_dKdiag, X):
        self.variance.gradient = np.sum(dL_dKdiag)

    def _install_gradients(self, g):
        self.variance.gradient = g[0]

    def gradients_X(self, dL_dK, X, X2=None):
        return np.zeros(np.asarray(X).shape)

    def gradients_X_diag(self, dL_dKdiag, X):
        """(reference `static.py:40-41`)"""
        return np.zeros(np.asarray(X).shape)

    def to_dict(self):
        return {"class": self._gpy_class, "name": self.name, "input_dim": self.input_dim,
                "active_dims": self.active_dims.tolist(), "variance": self.variance.values.tolist()}

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])


class White(Static):
    """(reference `static.py:63-98`): variance on the diagonal of K(X), zero cross-covariance."""
    kind = "white"
    _gpy_class = "GPy.kern.White"

    def K(self, X, X2=None):
        n = np.asarray(X).shape[0]
        return np.eye(n) * float(self.variance.values[0]) if X2 is None else np.zeros((n, np.asarray(X2).shape[0]))

    def update_gradients_full(self, dL_dK, X, X2=None):
        self.variance.gradient = np.trace(np.asarray(dL_dK)) if X2 is None else 0.


class Bias(Static):
    """(reference `static.py:151-173`): constant covariance."""
    kind = "bias"
    _gpy_class = "GPy.kern.Bias"

    def K(self, X, X2=None):
        n = np.asarray(X).shape[0]
        return np.full((n, n if X2 is None else np.asarray(X2).shape[0]), float(self.variance.values[0]))

    def update_gradients_full(self, dL_dK, X, X2=None):
        self.variance.gradient = np.sum(np.asarray(dL_dK))


class CombinationKernel(Parameterized):
    """Common part of `Add` and `Prod` (reference `GPy/kern/src/kern.py:327-390`): owns the parts, spans their input
    columns, and describes itself to the C-ABI as a list of parts with term ids (`mi355gp_part`)."""

    def __init__(self, parts, name):
        super(CombinationKernel, self).__init__(name)
        self.parts = parts
        self.input_dim = max(int(p.active_dims.max()) + 1 for p in parts)
        self.active_dims = np.arange(self.input_dim)
        self.device = parts[0].device
        for p in parts:
            self.link_parameter(p)

    def gradients_X_diag(self, dL_dKdiag, X):
        """Stationary and static leaves have a constant diagonal (reference `stationary.py:360-361`, `static.py:40-41`), and so
        has every sum / product of them (`add.py:102-106`, `prod.py:123-128`)."""
        return np.zeros(np.asarray(X).shape)

    def leaves(self):
        """the stationary / static kernels of the expression in link (= parameter, = gradient) order"""
        out = []
        for p in self.parts:
            out.extend(p.leaves() if isinstance(p, CombinationKernel) else [p])
        return out

    def _slice_X(self, X):
        return _lib.f64(np.asarray(X))

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])

    def _install_fused(self, g):
        i = 0
        for p in self.leaves():
            k = p._theta().size
            p._install_gradients(g[i:i + k])
            i += k

    def update_gradients_diag(self, dL_dKdiag, X):
        raise NotImplementedError

    def diag_variance(self):
        """Kdiag of the expression (a constant for stationary / static leaves)"""
        return float(self.Kdiag(np.zeros((1, self.input_dim)))[0])


class Add(CombinationKernel):
    """Sum of kernels (reference `GPy/kern/src/add.py:12-100`).  With `gpy_amd.ExactGaussianInference` the sum is
    assembled and differentiated on the device in the same fused call as a single kernel (C-ABI
    `mi355gp_exact_inference_sum`); parameter / gradient order = the parts' link order.  Parts may be `Prod`s."""

    def __init__(self, parts, name="sum"):
        flat = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, Add) else [p])          # add.py:24-33 flattens nested sums
        assert all(isinstance(p, (Stationary, Static, Prod)) for p in flat), \
            "Add supports stationary, White, Bias and Prod parts"
        super(Add, self).__init__(flat, name)

    def part_specs(self):
        """[(kind, ARD, theta, active_dims, term)] for the C-ABI (active_dims index the columns of the model's X);
        the factors of a `Prod` part share a non-zero term id."""
        specs, term = [], 0
        for p in self.parts:
            if isinstance(p, Prod):
                term += 1
                specs.extend((f.kind, f.ARD, f._theta(), f.active_dims, term) for f in p.parts)
            else:
                specs.append((p.kind, p.ARD, p._theta(), p.active_dims, 0))
        return specs

    def K(self, X, X2=None):
        out = None
        for p in self.parts:
            Kp = p.K(X, X2)
            out = Kp if out is None else out + Kp
        return out

    def Kdiag(self, X):
        return sum(p.Kdiag(X) for p in self.parts)

    def update_gradients_full(self, dL_dK, X, X2=None):
        """(reference `add.py:81-82`).  A device-resident dL_dK of the fused sum call carries every part's gradient."""
        if isinstance(dL_dK, DeviceResult) and X2 is None and dL_dK.matches_kernel(self):
            self._install_fused(dL_dK.fused_dtheta)
            return
        G = np.asarray(dL_dK)
        for p in self.parts:
            p.update_gradients_full(G, X, X2)

    def update_gradients_diag(self, dL_dKdiag, X):
        for p in self.parts:
            p.update_gradients_diag(dL_dKdiag, X)

    def gradients_X(self, dL_dK, X, X2=None):
        G = np.asarray(dL_dK)
        return sum(p.gradients_X(G, X, X2) for p in self.parts)

    def to_dict(self):
        return {"class": "GPy.kern.Add", "name": self.name, "parts": [p.to_dict() for p in self.parts]}


class Prod(CombinationKernel):
    """Product of kernels (reference `GPy/kern/src/prod.py:24-99`; nested products are flattened, `:33-41`).  Fused on the
    device like `Add`: the factors share a term id of `mi355gp_part`, K is multiplied up factor by factor in the
    K-build kernel and each factor's gradient pass weights dL_dK by the other factors' covariances."""

    def __init__(self, kernels, name="mul"):
        flat = []
        for k in kernels:
            flat.extend(k.parts if isinstance(k, Prod) else [k])
        assert all(isinstance(k, (Stationary, Static)) for k in flat), "Prod supports stationary, White and Bias factors"
        super(Prod, self).__init__(flat, name)

    def part_specs(self):
        return [(f.kind, f.ARD, f._theta(), f.active_dims, 1) for f in self.parts]

    def K(self, X, X2=None):                                                     # prod.py:58-65
        out = None
        for p in self.parts:
            Kp = p.K(X, X2)
            out = Kp if out is None else out * Kp
        return out

    def Kdiag(self, X):                                                          # prod.py:67-71
        out = None
        for p in self.parts:
            d = p.Kdiag(X)
            out = d if out is None else out * d
        return out

    def update_gradients_full(self, dL_dK, X, X2=None):
        """(reference `prod.py:86-99`): every factor sees dL_dK times the product of the other factors."""
        if isinstance(dL_dK, DeviceResult) and X2 is None and dL_dK.matches_kernel(self):
            self._install_fused(dL_dK.fused_dtheta)
            return
        G = np.asarray(dL_dK)
        Ks = [p.K(X, X2) for p in self.parts]
        for i, p in enumerate(self.parts):
            W = G
            for j, Kj in enumerate(Ks):
                if j != i:
                    W = W * Kj
            p.update_gradients_full(W, X, X2)

    def update_gradients_diag(self, dL_dKdiag, X):                               # prod.py:101-111
        ds = [p.Kdiag(X) for p in self.parts]
        for i, p in enumerate(self.parts):
            w = np.asarray(dL_dKdiag, dtype=float)
            for j, d in enumerate(ds):
                if j != i:
                    w = w * d
            p.update_gradients_diag(w, X)

    def gradients_X(self, dL_dK, X, X2=None):                                    # prod.py:113-121
        G = np.asarray(dL_dK)
        Ks = [p.K(X, X2) for p in self.parts]
        out = 0.
        for i, p in enumerate(self.parts):
            W = G
            for j, Kj in enumerate(Ks):
                if j != i:
                    W = W * Kj
            out = out + p.gradients_X(W, X, X2)
        return out

    def to_dict(self):
        return {"class": "GPy.kern.Prod", "name": self.name, "parts": [p.to_dict() for p in self.parts]}


KERNEL_CLASSES = {"rbf": RBF, "expquad": ExpQuad, "matern52": Matern52, "matern32": Matern32, "exponential": Exponential,
                  "white": White, "bias": Bias}

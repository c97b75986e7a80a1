"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the exact-GP hot path (NOT part of the product).

A NumPy/SciPy restatement of the algorithm GPy runs for one `GP.parameters_changed`
(reference `GPy/core/gp.py:278-280`): kernel matrix -> Ky -> pdinv -> alpha -> log marginal
likelihood -> dL_dK -> kernel / noise gradients.  Every function cites the reference lines it follows.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product (`gpy_amd`) never does and fails loudly when its HIP library is missing.

Parity pin: this restatement is checked against (a) the reference's own unmodified files executed
through `oracle/ref_loader.py` (tests/test_oracle_vs_reference.py, runs wherever /root/reference
exists) and (b) `tests/golden/*.npz`, generated from the reference by `oracle/make_golden.py`
(travels to the GPU box).  The reference itself ships no golden values for this path.

The LAPACK/BLAS arithmetic (dpotrf, dtrtri, dpotri, dpotrs, dsyrk) lives in SciPy -> OpenBLAS, a
third-party dependency of the reference (`setup.py:147`, scipy>=1.3.0; here scipy 1.15.3 /
OpenBLAS 0.3.29), called exactly where the reference calls it (`GPy/util/linalg.py:58,114,125,142,227,316`).
"""
import ctypes
import os

import numpy as np
from scipy import linalg as sla
from scipy.linalg import blas, lapack

LOG_2_PI = np.log(2.0 * np.pi)
KINDS = ("rbf", "matern52", "matern32", "exponential")

_HERE = os.path.dirname(os.path.abspath(__file__))
_native = None


def _load_native():
    """Optional C helpers (oracle/oracle_native.c): the serial Q*N*M lengthscale loop and symmetrify,
    restating the reference's Cython fast paths (`stationary_cython.pyx:53-62`, `linalg_cython.pyx:9-18`)."""
    global _native
    if _native is None:
        p = os.path.join(_HERE, "_build", "liboracle_native.so")
        if os.path.exists(p):
            lib = ctypes.CDLL(p)
            dp = ctypes.POINTER(ctypes.c_double)
            lib.oracle_lengthscale_grads.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_long, dp, dp, dp, dp]
            lib.oracle_lengthscale_grads.restype = None
            lib.oracle_symmetrify.argtypes = [ctypes.c_long, dp, ctypes.c_int]
            lib.oracle_symmetrify.restype = None
            _native = lib
        else:
            _native = False
    return _native


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# ----------------------------------------------------------------------------- linear algebra
def symmetrify(A, upper=False):
    """In-place mirror of one triangle onto the other (reference `util/linalg.py:356-379`)."""
    nat = _load_native()
    if nat and (A.flags.c_contiguous or A.flags.f_contiguous):
        # for an F-ordered array the roles of the two triangles swap
        up = bool(upper) if A.flags.c_contiguous else (not bool(upper))
        nat.oracle_symmetrify(A.shape[0], _dptr(A), int(up))
        return
    iu = np.triu_indices_from(A, k=1)
    if upper:
        A.T[iu] = A[iu]
    else:
        A[iu] = A.T[iu]


def tdot(mat):
    """mat @ mat.T through BLAS dsyrk + mirror (reference `util/linalg.py:299-323`)."""
    mat = np.asfortranarray(mat)
    n = mat.shape[0]
    out = np.zeros((n, n))
    out = blas.dsyrk(alpha=1.0, a=mat, beta=0.0, c=out, overwrite_c=1, trans=0, lower=0)
    symmetrify(out, upper=True)
    return np.ascontiguousarray(out)


def jitchol(A, maxtries=5):
    """Cholesky with the reference's jitter ladder (`util/linalg.py:56-75`): on dpotrf failure,
    non-positive diagonal -> LinAlgError; else jitter = mean(diag)*1e-6, x10 per try, <= maxtries."""
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L
    d = np.diag(A)
    if np.any(d <= 0.0):
        raise sla.LinAlgError("not pd: non-positive diagonal elements")
    jitter = d.mean() * 1e-6
    tries = 1
    while tries <= maxtries and np.isfinite(jitter):
        try:
            return sla.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
        except Exception:
            jitter *= 10
        finally:
            tries += 1
    raise sla.LinAlgError("not positive definite, even with jitter.")


def pdinv(A):
    """(A^-1, L, L^-1, logdet) as the reference computes them (`util/linalg.py:193-214`):
    jitchol, 2*sum(log diag L), dtrtri (result unused by the caller but executed), dpotri + 2x symmetrify."""
    L = jitchol(A)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    Li = lapack.dtrtri(np.asfortranarray(L), lower=1)[0]           # util/linalg.py:217-227
    Ai, _ = lapack.dpotri(np.asfortranarray(L), lower=1)           # util/linalg.py:127-145
    symmetrify(Ai)
    symmetrify(Ai)
    return Ai, L, Li, logdet


def dpotrs(L, B):
    """Solve (L L^T) X = B (reference `util/linalg.py:116-125`)."""
    return lapack.dpotrs(np.asfortranarray(L), B, lower=1)[0]


# ----------------------------------------------------------------------------- stationary kernels
def _as_ls(lengthscale, D, ARD):
    ls = np.atleast_1d(np.asarray(lengthscale, dtype=float))
    if ARD:
        if ls.size == 1:
            ls = np.full(D, float(ls[0]))
        assert ls.size == D
    else:
        assert ls.size == 1
    return ls


def unscaled_dist(X, X2=None):
    """Euclidean distances via |x|^2+|x'|^2-2x.x' with diag:=0 (symmetric case) and clip>=0
    (reference `kern/src/stationary.py:130-148`)."""
    if X2 is None:
        s = np.sum(np.square(X), 1)
        r2 = -2.0 * tdot(X) + (s[:, None] + s[None, :])
        r2[np.diag_indices_from(r2)] = 0.0
        r2 = np.clip(r2, 0, np.inf)
        return np.sqrt(r2)
    s1 = np.sum(np.square(X), 1)
    s2 = np.sum(np.square(X2), 1)
    r2 = -2.0 * np.dot(X, X2.T) + (s1[:, None] + s2[None, :])
    r2 = np.clip(r2, 0, np.inf)
    return np.sqrt(r2)


def scaled_dist(X, X2, lengthscale, ARD):
    """r: ARD scales the inputs first, iso divides afterwards (reference `stationary.py:150-168`)."""
    ls = _as_ls(lengthscale, X.shape[1], ARD)
    if ARD:
        return unscaled_dist(X / ls, None if X2 is None else X2 / ls)
    return unscaled_dist(X, X2) / ls


def K_of_r(kind, r, variance):
    """Covariance as a function of r.  RBF `rbf.py:51-52`; Matern52 `stationary.py:585-586`;
    Matern32 `:488-489`; Exponential `:382-383`."""
    if kind == "rbf":
        return variance * np.exp(-0.5 * r ** 2)
    if kind == "matern52":
        return variance * (1 + np.sqrt(5.0) * r + 5.0 / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)
    if kind == "matern32":
        return variance * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    if kind == "exponential":
        return variance * np.exp(-r)
    raise ValueError(kind)


def dK_dr(kind, r, variance):
    """dK/dr.  RBF `rbf.py:177-178`; Matern52 `stationary.py:588-589`; Matern32 `:491-492`;
    Exponential `:385-386`."""
    if kind == "rbf":
        return -r * K_of_r(kind, r, variance)
    if kind == "matern52":
        return variance * (10.0 / 3 * r - 5.0 * r - 5.0 * np.sqrt(5.0) / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)
    if kind == "matern32":
        return -3.0 * variance * r * np.exp(-np.sqrt(3.0) * r)
    if kind == "exponential":
        return -K_of_r(kind, r, variance)
    raise ValueError(kind)


def kern_K(kind, X, X2, variance, lengthscale, ARD):
    """`Stationary.K` (reference `stationary.py:105-115`)."""
    return K_of_r(kind, scaled_dist(X, X2, lengthscale, ARD), float(variance))


def kern_Kdiag(X, variance):
    """`Stationary.Kdiag` (reference `stationary.py:170-173`)."""
    out = np.empty(X.shape[0])
    out[:] = variance
    return out


def lengthscale_grads(tmp, X, X2):
    """grad[q] = sum_{n,m} tmp[n,m] (X[n,q]-X2[m,q])^2  (reference `stationary_cython.pyx:53-62`,
    NumPy fallback `stationary.py:234-235`)."""
    nat = _load_native()
    N, M = tmp.shape
    Q = X.shape[1]
    if nat:
        tmp = np.ascontiguousarray(tmp)
        Xc = np.ascontiguousarray(X, dtype=float)
        X2c = np.ascontiguousarray(X2, dtype=float)
        g = np.zeros(Q)
        nat.oracle_lengthscale_grads(N, M, Q, _dptr(tmp), _dptr(Xc), _dptr(X2c), _dptr(g))
        return g
    return np.array([np.sum(tmp * np.square(X[:, q:q + 1] - X2[:, q:q + 1].T)) for q in range(Q)])


def update_gradients_full(kind, dL_dK, X, X2, variance, lengthscale, ARD, K=None, r=None):
    """(dL/dvariance, dL/dlengthscale) from dL_dK (reference `stationary.py:193-213,225-243`).
    `K`/`r` may be passed to mimic paramz's `Cache_this` reuse of K and _scaled_dist between the
    inference step and the gradient step."""
    variance = float(variance)
    ls = _as_ls(lengthscale, X.shape[1], ARD)
    if r is None:
        r = scaled_dist(X, X2, ls, ARD)
    if K is None:
        K = K_of_r(kind, r, variance)
    dvar = np.sum(K * dL_dK) / variance
    dL_dr = dK_dr(kind, r, variance) * dL_dK
    if ARD:
        inv = 1.0 / np.where(r != 0.0, r, np.inf)                  # stationary.py:225-232
        tmp = dL_dr * inv
        X2_ = X if X2 is None else X2
        dlen = -lengthscale_grads(tmp, X, X2_) / ls ** 3
    else:
        dlen = np.atleast_1d(-np.sum(dL_dr * r) / ls[0])
    return dvar, dlen


# ----------------------------------------------------------------------------- exact inference
def exact_inference(K, Y, noise, mean=None, Z_tilde=None):
    """`ExactGaussianInference.inference` given K (reference `exact_gaussian_inference.py:42-74`).
    `noise` scalar or length-N vector.  Returns dict(L, alpha, lml, dL_dK, Wi, logdet, dL_dnoise)."""
    R = Y if mean is None else Y - mean
    Ky = K.copy()
    n = Ky.shape[0]
    Ky[np.arange(n), np.arange(n)] += np.asarray(noise, dtype=float) + 1e-8   # util/diag.py:85-98
    Wi, L, _Li, logdet = pdinv(Ky)
    alpha = dpotrs(L, R)
    lml = 0.5 * (-Y.size * LOG_2_PI - Y.shape[1] * logdet - np.sum(alpha * R))
    if Z_tilde is not None:
        lml += Z_tilde
    dL_dK = 0.5 * (tdot(alpha) - Y.shape[1] * Wi)
    # Gaussian.exact_inference_gradients = sum(diag(dL_dK)) (reference likelihoods/gaussian.py:78-79)
    return dict(L=L, alpha=alpha, lml=float(lml), dL_dK=dL_dK, Wi=Wi, logdet=float(logdet),
                dL_dnoise=float(np.sum(np.diag(dL_dK))), diag_dL_dK=np.diag(dL_dK).copy())


def parameters_changed(kind, X, Y, variance, lengthscale, ARD, noise, cached=True):
    """One full objective+gradient evaluation, the sequence `core/gp.py:278-280` runs:
    inference -> likelihood.update_gradients -> kern.update_gradients_full.
    `cached=True` reuses K and r in the gradient step like the real paramz `Cache_this(limit=3)` does."""
    ls = _as_ls(lengthscale, X.shape[1], ARD)
    r = scaled_dist(X, None, ls, ARD)
    K = K_of_r(kind, r, float(variance))
    res = exact_inference(K, Y, noise)
    if cached:
        dvar, dlen = update_gradients_full(kind, res["dL_dK"], X, None, variance, ls, ARD, K=K, r=r)
    else:
        dvar, dlen = update_gradients_full(kind, res["dL_dK"], X, None, variance, ls, ARD)
    res.update(K=K, dvar=float(dvar), dlen=np.asarray(dlen, dtype=float))
    return res


def predict(kind, X, Xnew, L, alpha, variance, lengthscale, ARD, noise=None, full_cov=False):
    """`PosteriorExact._raw_predict` (reference `posterior.py:273-302`) + optional Gaussian noise
    (`likelihoods/gaussian.py:102-110`)."""
    Kx = kern_K(kind, X, Xnew, variance, lengthscale, ARD)
    mu = Kx.T @ alpha
    tmp = lapack.dtrtrs(np.asfortranarray(L), Kx, lower=1)[0]
    if full_cov:
        var = kern_K(kind, Xnew, None, variance, lengthscale, ARD) - tmp.T @ tmp
        if noise is not None:
            var = var + np.eye(var.shape[0]) * noise
    else:
        var = (kern_Kdiag(Xnew, variance) - np.sum(np.square(tmp), 0))[:, None]
        if noise is not None:
            var = var + noise
    return mu, var


# ----------------------------------------------------------------------------- prediction-side callers (core/gp.py)
def gradients_X(kind, dL_dK, X, X2, variance, lengthscale, ARD):
    """`Stationary._gradients_X_pure` (stationary.py:330-346): derivative w.r.t. the rows of X."""
    ls = _as_ls(lengthscale, X.shape[1], ARD)
    r = scaled_dist(X, X2, ls, ARD)
    inv = 1.0 / np.where(r != 0.0, r, np.inf)
    tmp = inv * dK_dr(kind, r, float(variance)) * dL_dK
    if X2 is None:
        tmp = tmp + tmp.T
        X2 = X
    grad = np.empty(X.shape)
    for q in range(X.shape[1]):
        grad[:, q] = np.sum(tmp * (X[:, q][:, None] - X2[:, q][None, :]), axis=1)
    lsq = ls if ARD else np.full(X.shape[1], ls[0])
    return grad / lsq ** 2


def predictive_gradients(kind, X, Xnew, alpha, Wi, variance, lengthscale, ARD):
    """`GP.predictive_gradients` (core/gp.py:440-474, woodbury_inv.ndim == 2, no normaliser): (mean_jac (N*, Q, D),
    var_jac (N*, Q)); `gradients_X_diag` of a stationary kernel is zero (stationary.py:360-361)."""
    M, Q = Xnew.shape
    mean_jac = np.empty((M, Q, alpha.shape[1]))
    for i in range(alpha.shape[1]):
        mean_jac[:, :, i] = gradients_X(kind, alpha[:, i:i + 1].T, Xnew, X, variance, lengthscale, ARD)
    a2 = -2.0 * np.dot(kern_K(kind, Xnew, X, variance, lengthscale, ARD), Wi)
    return mean_jac, gradients_X(kind, a2, Xnew, X, variance, lengthscale, ARD)


def predictive_quantiles(mu, var, noise, quantiles):
    """`Gaussian.predictive_quantiles` (likelihoods/gaussian.py:118-119)"""
    from scipy import stats
    return [stats.norm.ppf(q / 100.) * np.sqrt(var + noise) + mu for q in quantiles]


def log_predictive_density(y_test, mu_star, var_star, noise):
    """`Gaussian.log_predictive_density` (likelihoods/gaussian.py:329-334)"""
    v = var_star + noise
    return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(v) - 0.5 * np.square(y_test - mu_star) / v


# ----------------------------------------------------------------------------- synthetic workload
def synthetic(N, D, seed=0, Dy=1):
    """SURVEY 8(d) synthetic inputs: X~N(0,1), Y = sin(x0)+0.5cos(2 x1)+0.1 eps."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    f = np.sin(X[:, 0]) + (0.5 * np.cos(2 * X[:, 1]) if D >= 2 else 0.0)
    Y = (f + 0.1 * rng.standard_normal(N))[:, None]
    if Dy > 1:
        Y = np.hstack([Y + 0.3 * j * np.cos(X[:, :1] * (j + 1)) for j in range(Dy)])
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)


def default_theta(D, ARD):
    """sigma^2=1.3; iso l=0.7 sqrt(D); ARD l_q=linspace(0.5,2,Q) sqrt(D/8); sigma_n^2=0.1."""
    if ARD:
        ls = np.linspace(0.5, 2.0, D) * np.sqrt(D / 8.0)
    else:
        ls = np.array([0.7 * np.sqrt(D)])
    return 1.3, ls, 0.1


# ----------------------------------------------------------------------------- sum kernels (GPy.kern.Add)
def _part_K(part, X, X2=None):
    """K of one part = (kind, ARD, variance, lengthscale, active_dims[, term]); 'white' / 'bias' follow
    `kern/src/static.py:77-81,165-167`; `active_dims` slices X (`kern.py:112-117`)."""
    kind, ARD, var, ls, dims = part[:5]
    n, m = X.shape[0], (X if X2 is None else X2).shape[0]
    if kind == "white":
        return np.eye(n) * var if X2 is None else np.zeros((n, m))
    if kind == "bias":
        return np.full((n, m), float(var))
    return kern_K(kind, X[:, dims], None if X2 is None else X2[:, dims], var, ls, ARD)


def _terms(parts):
    """[[part indices]] per summand: term id 0 (or a 5-tuple) = a summand of its own, parts sharing a non-zero
    term id are the factors of one `Prod` (the grouping of include/mi355gp.h `mi355gp_part.term`)."""
    groups, ids = [], []
    for i, p in enumerate(parts):
        t = p[5] if len(p) > 5 else 0
        if t != 0 and t in ids:
            groups[ids.index(t)].append(i)
        else:
            ids.append(t if t != 0 else None)
            groups.append([i])
    return groups


def sum_kern_K(parts, X, X2=None):
    """`Add.K` (reference `kern/src/add.py:58-72`) over summands that are single parts or `Prod`s of parts
    (`kern/src/prod.py:58-65`: element-wise product of the factors' K)."""
    n, m = X.shape[0], (X if X2 is None else X2).shape[0]
    K = np.zeros((n, m))
    for g in _terms(parts):
        T = _part_K(parts[g[0]], X, X2)
        for i in g[1:]:
            T = T * _part_K(parts[i], X, X2)
        K += T
    return K


def _part_grads(part, G, X):
    """`update_gradients_full(G, X)` of one part (White: trace, Bias: sum, `static.py:89-93,169-170`)."""
    kind, ARD, var, ls, dims = part[:5]
    if kind == "white":
        return np.array([np.trace(G)])
    if kind == "bias":
        return np.array([np.sum(G)])
    dv, dl = update_gradients_full(kind, G, X[:, dims], None, var, ls, ARD)
    return np.concatenate([[dv], np.atleast_1d(dl)])


def sum_parameters_changed(parts, X, Y, noise):
    """One `GP.parameters_changed` with a sum(-of-products) kernel: inference on K, then every part's
    `update_gradients_full` (`add.py:81-82`); a factor of a `Prod` receives dL_dK times the product of the other
    factors' K (`prod.py:377-385`).  dtheta is concatenated in part order."""
    K = sum_kern_K(parts, X)
    res = exact_inference(K, Y, noise)
    grads = [None] * len(parts)
    for g in _terms(parts):
        for i in g:
            G = res["dL_dK"]
            for j in g:
                if j != i:
                    G = G * _part_K(parts[j], X)
            grads[i] = _part_grads(parts[i], G, X)
    res.update(K=K, dtheta=np.concatenate(grads))
    return res


def sum_kdiag(parts):
    """Kdiag of the expression: sum over summands of the product of variances (`add.py:74-79`, `prod.py:67-71`)."""
    return sum(float(np.prod([parts[i][2] for i in g])) for g in _terms(parts))


def sum_predict(parts, X, Xs, L, alpha):
    """`PosteriorExact._raw_predict` (posterior.py:273-302) with a sum(-of-products) kernel."""
    Kx = sum_kern_K(parts, X, Xs)
    mu = Kx.T @ alpha
    tmp = lapack.dtrtrs(np.asfortranarray(L), np.asfortranarray(Kx), lower=1)[0]
    kdiag = sum_kdiag(parts)
    var = (kdiag - np.sum(tmp * tmp, 0))[:, None]
    cov = sum_kern_K(parts, Xs) - tmp.T @ tmp
    return mu, var, cov


def studentt_inference(K, Y, nu):
    """`ExactStudentTInference.inference` given K (reference `exact_studentt_inference.py:20-52`)."""
    from scipy.special import digamma, gammaln
    Ky = K.copy()
    n, D = Y.shape
    Ky[np.arange(n), np.arange(n)] += 1e-8
    Wi, L, _Li, logdet = pdinv(Ky)
    alpha = dpotrs(L, Y)
    beta = float(np.sum(alpha * Y))
    lml = 0.5 * (-n * np.log((nu - 2) * np.pi) - logdet - (nu + n) * np.log(1 + beta / (nu - 2)))
    lml += gammaln((nu + n) / 2) - gammaln(nu / 2)
    dL_dK = 0.5 * ((nu + n) / (nu + beta - 2) * tdot(alpha) - D * Wi)
    dL_dnu = -n / (nu - 2.0) + digamma(0.5 * (nu + n)) - digamma(0.5 * nu)
    dL_dnu -= np.log(1 + beta / (nu - 2.0))
    dL_dnu += ((nu + n) * beta) / ((nu - 2) * (beta + nu - 2))
    dL_dnu *= 0.5
    return dict(L=L, alpha=alpha, lml=float(lml), dL_dK=dL_dK, dL_dnu=float(dL_dnu),
                dL_dm=(nu + n) / (nu + beta - 2) * alpha, beta=beta)

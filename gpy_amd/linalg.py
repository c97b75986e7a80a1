"""Device-backed mirror of the two N^3 entry points of `GPy/util/linalg.py` that the exact path (and every other inference
method of the reference: Laplace, EP, VarDTC) goes through:

    jitchol(A, maxtries=5)  -> L                      (reference `util/linalg.py:56-75`)
    pdinv(A)                -> (Ai, L, Li, logdet)    (reference `util/linalg.py:193-214`)

The matrix travels to the device once, the factorisation (`mi355gp_potrf`) / factorisation + inverse (`mi355gp_pdinv_full`:
blocked Cholesky, triangular inverse, X^T X) run on the device, the results come back as NumPy arrays in the reference's
layout (C-order `Ai`, lower-triangular `L` with a zero upper triangle, `Li = L^-1`).  Errors are the reference's:
`numpy.linalg.LinAlgError("not pd: non-positive diagonal elements")` when the diagonal itself is not positive, otherwise
the jitter ladder (mean(diag) * 1e-6, times 10 per attempt) and `LinAlgError("not positive definite, even with jitter.")`.
No CPU fallback: without the device the calls raise.
"""
import logging

import numpy as np

from . import _lib

LinAlgError = np.linalg.LinAlgError


def _ladder(A, maxtries, attempt):
    """`attempt(A_jittered)` -> (result, info); the reference's jitter ladder around it (`util/linalg.py:61-75`)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    res, info = attempt(A)
    if info == 0:
        return res
    diagA = np.diag(A)
    if np.any(diagA <= 0.):
        raise LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        res, info = attempt(A + np.eye(A.shape[0]) * jitter)
        if info == 0:
            logging.getLogger(__name__).warning("Added jitter of {:.10e}".format(jitter))
            return res
        jitter *= 10
        num_tries += 1
    raise LinAlgError("not positive definite, even with jitter.")


def jitchol(A, maxtries=5, device=0):
    """Lower Cholesky factor with the reference's jitter ladder (`util/linalg.py:56-75`), factorised on the device."""
    def attempt(M):
        L, info, _ = _lib.potrf(M, device=device)
        return L, info
    return _ladder(A, maxtries, attempt)


def pdinv(A, *args, **kw):
    """(Ai, L, Li, logdet) of a positive-definite matrix (`util/linalg.py:193-214`), everything N^3 on the device."""
    device = kw.pop("device", 0)

    def attempt(M):
        Ai, L, Li, logdet, info = _lib.pdinv_full(M, device=device)
        return (Ai, L, Li, logdet), info
    return _ladder(A, kw.pop("maxtries", 5), attempt)

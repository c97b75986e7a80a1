"""TEST INFRASTRUCTURE ONLY -- golden vectors AT THE FULL BASELINE.json SIZES, generated from the reference's own code.

    python oracle/make_golden_baseline.py [c2] [c3] [c3s] [c4] [c5]        (default: all; ~25 min on 8 cores, <= 45 GB)

    c2   RBF iso,            N=4096,   D=8          reference through oracle/ref_loader.py (Cython ext built)
    c3   Matern-5/2 ARD,     N=16384,  D=32         reference through oracle/ref_loader.py
    c3s  Matern-5/2 ARD D=32 at N=6144 (first size on the overlapped-inverse schedule)     reference
    c4   RBF iso,            N=32768,  D=8          LEAN oracle below: the reference's pdinv keeps ~8 live N x N
                                                    temporaries (8.6 GB each) and does not fit this container's 62 GB
    c5   VarDTC RBF iso,     N=200000, M=2048, D=16 reference's VarDTC + SparseGP._update_gradients (ref_loader)

Inputs are NOT stored: tests regenerate them from the seed with oracle.gp_oracle.synthetic (NumPy's
default_rng stream is stable and the GPU box runs the same image).  Stored: LML, logdet, alpha, all gradients,
diag(dL_dK), a few rows of Ky^-1 / L -- < 2 MB in total.

The lean oracle (`lean_exact`) makes the same LAPACK calls the reference makes for this path
(dpotrf / dpotrs / dpotri: util/linalg.py:58,125,142) in place on ONE N x N buffer, skips the dtrtri whose result
the reference never uses (util/linalg.py:204), and evaluates the gradient sums of stationary.py:199,212-213 row block
by row block.  tests/test_oracle_baseline.py pins it against oracle.gp_oracle (itself pinned against the reference)
at a size both can run.
"""
import importlib
import os
import sys
import time

import numpy as np
from scipy.linalg import lapack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref_cython, ref_loader  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from oracle.sparse_oracle import synthetic_Z  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ROWS = 6          # sampled rows of Ky^-1 and L


def sample_rows(N, seed=5):
    return np.sort(np.random.default_rng(seed).choice(N, ROWS, replace=False))


def exact_from_reference(ns, kind, N, D, ARD, seed=0):
    X, Y = O.synthetic(N, D, seed=seed)
    var, ls, noise = O.default_theta(D, ARD)
    k = ref_loader.make_kernel(ns, kind, D, var, ls if ARD else float(ls[0]), ARD)
    lik = ns.Gaussian(variance=noise)
    t0 = time.time()
    post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    dt = time.time() - t0
    rows = sample_rows(N)
    G = np.asarray(gd["dL_dK"])
    alpha = np.asarray(post.woodbury_vector)
    Wi_rows = (alpha[rows] @ alpha.T - 2.0 * G[rows]) / Y.shape[1]          # dL_dK = 0.5 (aa^T - Dy Wi)
    L = np.asarray(post.woodbury_chol)
    out = dict(kind=kind, ARD=ARD, N=N, D=D, seed=seed, variance=var, lengthscale=ls, noise=noise,
               lml=float(lml), alpha=alpha, dvar=np.atleast_1d(np.asarray(k.variance.gradient, float)).copy(),
               dlen=np.atleast_1d(np.asarray(k.lengthscale.gradient, float)).copy(),
               dnoise=np.atleast_1d(np.asarray(lik.variance.gradient, float)).copy(),
               diag_dL_dK=np.diag(G).copy(), rows=rows, Wi_rows=Wi_rows, L_rows=L[rows].copy(),
               logdet=float(2.0 * np.sum(np.log(np.diag(L)))), source="reference", seconds=dt)
    return out


def lean_exact(kind, X, Y, variance, lengthscale, noise, block=2048):
    """Isotropic stationary kernel, Dy columns; returns dict(lml, logdet, alpha, dvar, dlen, dnoise, diag_dL_dK, A)
    where A holds Ky^-1 in its lower triangle (row-major)."""
    N, D = X.shape
    Dy = Y.shape[1]
    ell = float(np.atleast_1d(lengthscale)[0])
    variance = float(variance)
    s = np.sum(np.square(X), 1)

    def r_block(i0, i1):                                       # stationary.py:130-168 (iso: divide r by l afterwards)
        r2 = -2.0 * (X[i0:i1] @ X.T) + (s[i0:i1, None] + s[None, :])
        r2[np.arange(i1 - i0), np.arange(i0, i1)] = 0.0
        np.clip(r2, 0, np.inf, out=r2)
        return np.sqrt(r2) / ell

    A = np.empty((N, N))                                       # K -> Ky -> L -> Ky^-1, all in this one buffer
    for i0 in range(0, N, block):
        i1 = min(N, i0 + block)
        A[i0:i1] = O.K_of_r(kind, r_block(i0, i1), variance)
    A[np.arange(N), np.arange(N)] += noise + 1e-8              # exact_gaussian_inference.py:55-56
    # A is symmetric, so its transpose view is the F-ordered matrix LAPACK wants: no N x N copy
    c, info = lapack.dpotrf(A.T, lower=1, overwrite_a=1, clean=1)          # util/linalg.py:58
    assert info == 0 and np.shares_memory(c, A)
    # c = A.T holds L (lower, column-major)  <=>  A holds L^T (upper, row-major)
    logdet = 2.0 * np.sum(np.log(np.diag(A)))
    alpha = lapack.dpotrs(c, Y, lower=1)[0]                    # util/linalg.py:116-125
    L_rows_idx = sample_rows(N)
    L_rows = np.ascontiguousarray(c[L_rows_idx])               # rows of L
    ci, info = lapack.dpotri(c, lower=1, overwrite_c=1)        # util/linalg.py:127-145 -> lower triangle of Ky^-1 (F-order)
    assert info == 0 and np.shares_memory(ci, A)
    # ci lower (F) = A upper (C): Wi[i, j] for j >= i is A[i, j]
    lml = 0.5 * (-Y.size * O.LOG_2_PI - Dy * logdet - np.sum(alpha * Y))
    dvar = dlr = 0.0
    diagG = np.empty(N)
    Wi_rows = np.empty((ROWS, N))
    for i0 in range(0, N, block):
        i1 = min(N, i0 + block)
        Wi = np.empty((i1 - i0, N))
        Wi[:, i0:] = A[i0:i1, i0:]                             # j >= i0: upper part as stored (fix the in-block lower below)
        Wi[:, :i0] = A[:i0, i0:i1].T
        blk = Wi[:, i0:i1]
        iu = np.triu_indices(i1 - i0, 1)
        blk.T[iu] = blk[iu]                                    # mirror the diagonal block's upper triangle down
        G = 0.5 * (alpha[i0:i1] @ alpha.T - Dy * Wi)           # exact_gaussian_inference.py:70
        r = r_block(i0, i1)
        K = O.K_of_r(kind, r, variance)
        dvar += np.sum(K * G) / variance                       # stationary.py:199
        dlr += np.sum(O.dK_dr(kind, r, variance) * G * r)      # stationary.py:202,212-213
        diagG[i0:i1] = G[np.arange(i1 - i0), np.arange(i0, i1)]
        for k, ri in enumerate(L_rows_idx):
            if i0 <= ri < i1:
                Wi_rows[k] = Wi[ri - i0]
    return dict(lml=float(lml), logdet=float(logdet), alpha=alpha, dvar=np.array([dvar]), dlen=np.array([-dlr / ell]),
                dnoise=np.array([np.sum(diagG)]), diag_dL_dK=diagG, rows=L_rows_idx, Wi_rows=Wi_rows, L_rows=L_rows)


def exact_lean(kind, N, D, seed=0):
    X, Y = O.synthetic(N, D, seed=seed)
    var, ls, noise = O.default_theta(D, False)
    t0 = time.time()
    r = lean_exact(kind, X, Y, var, ls, noise)
    r.update(kind=kind, ARD=False, N=N, D=D, seed=seed, variance=var, lengthscale=ls, noise=noise,
             source="lean oracle (dpotrf/dpotrs/dpotri in place)", seconds=time.time() - t0)
    return r


def sparse_from_reference(ns, N, M, D, seed=0):
    from oracle.make_golden_sparse import run_reference
    X, Y = O.synthetic(N, D, seed=seed)
    Z = synthetic_Z(X, M, seed)
    var, ls, noise = O.default_theta(D, False)
    t0 = time.time()
    r = run_reference(ns, "rbf", X, Z, Y, var, ls, False, noise)
    rows = sample_rows(M)
    out = dict(kind="rbf", ARD=False, N=N, M=M, D=D, seed=seed, variance=var, lengthscale=ls, noise=noise,
               lml=r["lml"], dtheta=r["dtheta"], dnoise=r["dnoise"], dZ=r["dZ"], woodbury_vector=r["woodbury_vector"],
               rows=rows, woodbury_inv_rows=r["woodbury_inv"][rows], dL_dKmm_rows=r["dL_dKmm"][rows],
               source="reference", seconds=time.time() - t0)
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("%-34s lml=% .12e  %.1f s  (%s)" % (name, d["lml"], d["seconds"], d["source"]), flush=True)


def main(argv):
    which = set(argv) or {"c2", "c3", "c3s", "c4", "c5"}
    build_ref_cython.build()
    ns = ref_loader.load()
    assert ns.use_stationary_cython and ns.use_linalg_cython, "reference Cython extensions not picked up"
    if "c2" in which:
        save("baseline_c2_rbf_n4096_d8", exact_from_reference(ns, "rbf", 4096, 8, False))
    if "c3s" in which:
        save("baseline_c3s_matern52_ard_n6144_d32", exact_from_reference(ns, "matern52", 6144, 32, True))
    if "c3" in which:
        save("baseline_c3_matern52_ard_n16384_d32", exact_from_reference(ns, "matern52", 16384, 32, True))
    if "c5" in which:
        save("baseline_c5_sparse_rbf_n200000_m2048_d16", sparse_from_reference(ns, 200000, 2048, 16))
    if "c4" in which:
        save("baseline_c4_rbf_n32768_d8", exact_lean("rbf", 32768, 8))


if __name__ == "__main__":
    main(sys.argv[1:])

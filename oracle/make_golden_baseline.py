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

The lean oracle (`lean_exact`) makes the LAPACK calls the reference makes for this path (dpotrf / dtrtrs / dtrtri /
dpotri: util/linalg.py:58,114,125,142,227) on the blocks of a 2 x 2 partition of Ky (a monolithic dpotrf of a
32768 x 32768 matrix segfaults in this SciPy/OpenBLAS build -- the reference's own jitchol would too) and evaluates the
gradient sums of stationary.py:199,212-213 row block by row block.  tests/test_oracle_baseline.py pins it against oracle.gp_oracle (itself pinned against the reference)
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


def lean_exact(kind, X, Y, variance, lengthscale, noise, block=2048, n1=None):
    """Isotropic stationary kernel, Dy columns; returns dict(lml, logdet, alpha, dvar, dlen, dnoise, diag_dL_dK, rows, Wi_rows,
    L_rows).  The matrix is held as the 2 x 2 block partition [[A11, .], [A21, A22]] (n1 = N // 2 rows in the first block)
    and factorised / inverted block-wise with the LAPACK calls of the reference's path on the blocks -- what LAPACK's own
    blocked dpotrf / dpotri do internally, one level up.  (A monolithic lapack.dpotrf segfaults at N = 32768 in this
    SciPy/OpenBLAS build -- the reference's own jitchol would as well -- and the 2 x 2 form keeps every call below 2^29
    elements.)"""
    N, D = X.shape
    Dy = Y.shape[1]
    ell = float(np.atleast_1d(lengthscale)[0])
    variance = float(variance)
    s = np.sum(np.square(X), 1)
    n1 = N // 2 if n1 is None else n1
    n2 = N - n1

    def r_block(i0, i1, j0=0, j1=None):                        # stationary.py:130-168 (iso: divide r by l afterwards)
        j1 = N if j1 is None else j1
        r2 = -2.0 * (X[i0:i1] @ X[j0:j1].T) + (s[i0:i1, None] + s[None, j0:j1])
        lo, hi = max(i0, j0), min(i1, j1)
        if hi > lo:
            r2[np.arange(lo, hi) - i0, np.arange(lo, hi) - j0] = 0.0
        np.clip(r2, 0, np.inf, out=r2)
        return np.sqrt(r2) / ell

    def build(i0, i1, j0, j1):                                 # F-ordered block of Ky
        B = np.empty((i1 - i0, j1 - j0), order="F")
        for a in range(i0, i1, block):
            b = min(i1, a + block)
            B[a - i0:b - i0] = O.K_of_r(kind, r_block(a, b, j0, j1), variance)
        if i0 == j0:
            B[np.arange(i1 - i0), np.arange(i1 - i0)] += noise + 1e-8          # exact_gaussian_inference.py:55-56
        return B

    A11, A21, A22 = build(0, n1, 0, n1), build(n1, N, 0, n1), build(n1, N, n1, N)
    L11, info = lapack.dpotrf(A11, lower=1, overwrite_a=1, clean=1)            # util/linalg.py:58
    assert info == 0
    del A11
    # L21 = A21 L11^-T  (dtrsm, side R, trans T): solve X L11^T = A21
    from scipy.linalg import blas
    L21 = blas.dtrsm(1.0, L11, A21, side=1, lower=1, trans_a=1, overwrite_b=1)
    del A21
    A22 = blas.dsyrk(-1.0, L21, beta=1.0, c=A22, lower=1, overwrite_c=1)       # S = A22 - L21 L21^T (lower)
    L22, info = lapack.dpotrf(A22, lower=1, overwrite_a=1, clean=1)
    assert info == 0
    del A22
    logdet = 2.0 * (np.sum(np.log(np.diag(L11))) + np.sum(np.log(np.diag(L22))))
    # alpha = Ky^-1 Y by forward / back substitution through the blocks (dpotrs, util/linalg.py:116-125)
    y1 = lapack.dtrtrs(L11, Y[:n1], lower=1)[0]
    y2 = lapack.dtrtrs(L22, Y[n1:] - L21 @ y1, lower=1)[0]
    a2 = lapack.dtrtrs(L22, y2, lower=1, trans=1)[0]
    a1 = lapack.dtrtrs(L11, y1 - L21.T @ a2, lower=1, trans=1)[0]
    alpha = np.vstack([a1, a2])
    rows = sample_rows(N)
    L_rows = np.zeros((ROWS, N))
    for k, ri in enumerate(rows):
        if ri < n1:
            L_rows[k, :n1] = L11[ri]
        else:
            L_rows[k, :n1], L_rows[k, n1:] = L21[ri - n1], L22[ri - n1]
    # Ky^-1 = W^T W with W = L^-1 = [[W11, 0], [W21, W22]], W21 = -W22 L21 W11   (dpotri, util/linalg.py:127-145)
    W11 = lapack.dtrtri(L11, lower=1)[0]
    W22 = lapack.dtrtri(L22, lower=1)[0]
    W21 = blas.dtrmm(-1.0, W22, blas.dtrmm(1.0, W11, L21, side=1, lower=1), side=0, lower=1)
    del L21
    K22 = lapack.dpotri(L22, lower=1, overwrite_c=1)[0]       # W22^T W22 (lower)
    K21 = blas.dtrmm(1.0, W22, W21, side=0, lower=1, trans_a=1)                # W22^T W21
    del W22, L22
    K11 = lapack.dpotri(L11, lower=1, overwrite_c=1)[0]       # W11^T W11 (lower)
    K11 = blas.dsyrk(1.0, W21, beta=1.0, c=K11, trans=1, lower=1, overwrite_c=1)   # + W21^T W21
    del W11, W21, L11
    O.symmetrify(K11)
    O.symmetrify(K22)
    lml = 0.5 * (-Y.size * O.LOG_2_PI - Dy * logdet - np.sum(alpha * Y))
    dvar = dlr = 0.0
    diagG = np.empty(N)
    Wi_rows = np.empty((ROWS, N))
    edges = sorted(set(range(0, N, block)) | {n1, N})           # row blocks never straddle the block-row boundary n1
    for i0, i1 in zip(edges[:-1], edges[1:]):
        Wi = np.empty((i1 - i0, N))
        if i0 < n1:
            Wi[:, :n1], Wi[:, n1:] = K11[i0:i1], K21[:, i0:i1].T
        else:
            Wi[:, :n1], Wi[:, n1:] = K21[i0 - n1:i1 - n1], K22[i0 - n1:i1 - n1]
        G = 0.5 * (alpha[i0:i1] @ alpha.T - Dy * Wi)           # exact_gaussian_inference.py:70
        r = r_block(i0, i1)
        K = O.K_of_r(kind, r, variance)
        dvar += np.sum(K * G) / variance                       # stationary.py:199
        dlr += np.sum(O.dK_dr(kind, r, variance) * G * r)      # stationary.py:202,212-213
        diagG[i0:i1] = G[np.arange(i1 - i0), np.arange(i0, i1)]
        for k, ri in enumerate(rows):
            if i0 <= ri < i1:
                Wi_rows[k] = Wi[ri - i0]
    return dict(lml=float(lml), logdet=float(logdet), alpha=alpha, dvar=np.array([dvar]), dlen=np.array([-dlr / ell]),
                dnoise=np.array([np.sum(diagG)]), diag_dL_dK=diagG, rows=rows, Wi_rows=Wi_rows, L_rows=L_rows)


def exact_lean(kind, N, D, seed=0):
    X, Y = O.synthetic(N, D, seed=seed)
    var, ls, noise = O.default_theta(D, False)
    t0 = time.time()
    r = lean_exact(kind, X, Y, var, ls, noise)
    r.update(kind=kind, ARD=False, N=N, D=D, seed=seed, variance=var, lengthscale=ls, noise=noise,
             source="lean oracle (2x2-blocked dpotrf/dtrtri/dpotri)", seconds=time.time() - t0)
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

"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the sparse (VarDTC) path; NOT part of the product.

NumPy/SciPy restatement of what `SparseGP.parameters_changed` runs for certain inputs and a homoscedastic
Gaussian likelihood (reference `GPy/core/sparse_gp.py:76-119`):
    VarDTC.inference                       `GPy/inference/latent_function_inference/var_dtc.py:66-215`
    _compute_dL_dpsi / _compute_dL_dR / _compute_log_marginal_likelihood      `var_dtc.py:217-276`
    kernel gradients  update_gradients_diag + update_gradients_full(dL_dKnm, X, Z) + (dL_dKmm, Z)
                                           `sparse_gp.py:108-115`, `stationary.py:175-213`
    inducing-input gradients  gradients_X(dL_dKmm, Z) + gradients_X(dL_dKnm.T, Z, X)
                                           `sparse_gp.py:116-118`, `stationary.py:245-252,330-358`
Pinned against the reference's own code through `oracle/ref_loader.py` (tests/test_oracle_vs_reference.py)
and `tests/golden/sparse_*.npz` (generator: `oracle/make_golden_sparse.py`).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy.linalg import lapack

from . import gp_oracle as O

CONST_JITTER = 1e-8        # var_dtc.py:24


def _dtrtrs(L, B, trans=0):
    """util/linalg.py:95-114"""
    return lapack.dtrtrs(np.asfortranarray(L), np.asfortranarray(B), lower=1, trans=trans)[0]


def backsub_both_sides(L, X):
    """L^-T X L^-1 (util/linalg.py:381-387, transpose='left')"""
    tmp = _dtrtrs(L, X, trans=1)
    return _dtrtrs(L, tmp.T, trans=1).T


def gradients_X(kind, dL_dK, X, X2, variance, lengthscale, ARD):
    """`Stationary._gradients_X_pure` (stationary.py:330-346): derivative w.r.t. the rows of X."""
    ls = O._as_ls(lengthscale, X.shape[1], ARD)
    r = O.scaled_dist(X, X2, ls, ARD)
    inv = 1.0 / np.where(r != 0.0, r, np.inf)
    tmp = inv * O.dK_dr(kind, r, float(variance)) * dL_dK
    if X2 is None:
        tmp = tmp + tmp.T
        X2 = X
    grad = np.empty(X.shape)
    for q in range(X.shape[1]):
        grad[:, q] = np.sum(tmp * (X[:, q][:, None] - X2[:, q][None, :]), axis=1)
    lsq = ls if ARD else np.full(X.shape[1], ls[0])
    return grad / lsq ** 2


def vardtc(kind, X, Z, Y, variance, lengthscale, ARD, noise_var):
    """One `SparseGP.parameters_changed`.  Returns dict(lml, dtheta=[dvar, dlen...], dnoise, dZ, woodbury_vector,
    woodbury_inv, dL_dKmm, Lm, Kmm, psi2, A, LB)."""
    N, Dy = Y.shape
    M = Z.shape[0]
    variance = float(variance)
    ls = O._as_ls(lengthscale, X.shape[1], ARD)
    beta = 1.0 / max(float(noise_var), CONST_JITTER)                       # var_dtc.py:78-80
    VVT = beta * Y                                                          # :88
    trYYT = float(np.sum(np.square(Y)))                                     # :89 (get_trYYT)
    Kmm = O.kern_K(kind, Z, None, variance, ls, ARD).copy()                 # :93
    Kmm[np.arange(M), np.arange(M)] += CONST_JITTER                         # :94
    Lm = O.jitchol(Kmm)                                                     # :95
    psi0 = np.full(N, variance)                                             # :125-126 (Kdiag)
    psi1 = O.kern_K(kind, X, Z, variance, ls, ARD)                          # :127-128
    tmp = _dtrtrs(Lm, (psi1 * np.sqrt(beta)).T)                             # :129-133
    A = tmp @ tmp.T                                                         # :134 (tdot)
    B = np.eye(M) + A                                                       # :137
    LB = O.jitchol(B)                                                       # :138
    tmp = _dtrtrs(Lm, psi1.T)                                               # :141
    LBi_Lmi_psi1 = _dtrtrs(LB, tmp)                                         # :142
    c = LBi_Lmi_psi1 @ VVT                                                  # :143  (_LBi_Lmi_psi1Vf)
    tmp = _dtrtrs(LB, c, trans=1)                                           # :144
    Cpsi1Vf = _dtrtrs(Lm, tmp, trans=1)                                     # :145
    delit = c @ c.T                                                         # :150
    data_fit = float(np.trace(delit))                                       # :151
    P = backsub_both_sides(LB, Dy * np.eye(M) + delit)                      # :152  (DBi_plus_BiPBi)
    dL_dKmm = backsub_both_sides(Lm, -0.5 * P - 0.5 * B * Dy + Dy * np.eye(M))   # :153-158
    # _compute_dL_dpsi (var_dtc.py:217-233), homoscedastic, certain inputs
    dL_dpsi0 = -0.5 * Dy * beta * np.ones(N)
    dL_dpsi1 = VVT @ Cpsi1Vf.T
    dL_dpsi2 = beta * 0.5 * backsub_both_sides(Lm, Dy * np.eye(M) - P)
    dL_dpsi1 = dL_dpsi1 + 2.0 * psi1 @ dL_dpsi2
    # _compute_log_marginal_likelihood (var_dtc.py:264-276)
    lik_1 = -0.5 * N * Dy * (np.log(2.0 * np.pi) - np.log(beta)) - 0.5 * beta * trYYT
    lik_2 = -0.5 * Dy * (np.sum(beta * psi0) - np.trace(A))
    lik_3 = -Dy * np.sum(np.log(np.diag(LB)))
    lml = lik_1 + lik_2 + lik_3 + 0.5 * data_fit
    # _compute_dL_dR (var_dtc.py:258-261)
    dL_dR = -0.5 * N * Dy * beta + 0.5 * trYYT * beta ** 2
    dL_dR += 0.5 * Dy * (psi0.sum() * beta ** 2 - np.trace(A) * beta)
    dL_dR += beta * (0.5 * np.sum(A * P) - data_fit)
    # posterior (var_dtc.py:198-214)
    Bi = -lapack.dpotri(np.asfortranarray(LB), lower=1)[0]
    Bi = np.tril(Bi) + np.tril(Bi, -1).T
    Bi[np.arange(M), np.arange(M)] += 1.0
    woodbury_inv = backsub_both_sides(Lm, Bi)
    # SparseGP._update_gradients (sparse_gp.py:108-118)
    dvar = float(np.sum(dL_dpsi0))                                          # update_gradients_diag (stationary.py:175-184)
    dlen = np.zeros(ls.size)
    dv, dl = O.update_gradients_full(kind, dL_dpsi1, X, Z, variance, ls, ARD)
    dvar += float(dv)
    dlen = dlen + np.asarray(dl, float)
    dv, dl = O.update_gradients_full(kind, dL_dKmm, Z, None, variance, ls, ARD)
    dvar += float(dv)
    dlen = dlen + np.asarray(dl, float)
    dZ = gradients_X(kind, dL_dKmm, Z, None, variance, ls, ARD) + gradients_X(kind, dL_dpsi1.T, Z, X, variance, ls, ARD)
    return dict(lml=float(lml), dtheta=np.concatenate([[dvar], dlen]), dnoise=float(dL_dR), dZ=dZ,
                woodbury_vector=Cpsi1Vf, woodbury_inv=woodbury_inv, dL_dKmm=dL_dKmm, Lm=Lm, Kmm=Kmm,
                psi2=psi1.T @ psi1, A=A, LB=LB, dL_dKnm=dL_dpsi1)


def vardtc_general(parts, X, Z, R, noise):
    """The same evaluation for a SUM of parts (`GPy.kern.Add` of stationary / White / Bias kernels, add.py:58-84;
    parts = [(kind, ARD, variance, lengthscale, active_dims[, term])] as in gp_oracle.sum_kern_K; parts sharing a non-zero
    term id are the factors of one `Prod`, prod.py:58-99: every factor sees dL_dK times the other factors' K, its
    gradients_X likewise, and its update_gradients_diag dL_dKdiag times the other factors' Kdiag), per-point noise variances
    (heteroscedastic precision, var_dtc.py:78-86,126-129,224-226,240-256,267-269; any Dy) and R = Y - mean_function.f(X)
    (var_dtc.py:73-76,88-89).  Returns dict(lml, dtheta (concatenated in part order), dnoise (scalar, or dL_dR: N-vector / N x Dy),
    dZ, woodbury_vector, woodbury_inv, dL_dKmm, dL_dKnm, dL_dm)."""
    N, Dy = R.shape
    M = Z.shape[0]
    noise = np.atleast_1d(np.asarray(noise, dtype=float)).ravel()
    het = noise.size > 1
    beta = 1.0 / np.fmax(noise, CONST_JITTER)                                # (1,) or (N,)
    bcol = beta[:, None] if het else beta[0]
    VVT = bcol * R
    Kmm = O.sum_kern_K(parts, Z).copy()
    Kmm[np.arange(M), np.arange(M)] += CONST_JITTER
    Lm = O.jitchol(Kmm)
    psi0 = np.full(N, O.sum_kdiag(parts))
    psi1 = O.sum_kern_K(parts, X, Z)
    tmp = _dtrtrs(Lm, (psi1 * np.sqrt(bcol)).T)
    A = tmp @ tmp.T
    B = np.eye(M) + A
    LB = O.jitchol(B)
    LBi_Lmi_psi1 = _dtrtrs(LB, _dtrtrs(Lm, psi1.T))
    c = LBi_Lmi_psi1 @ VVT
    Cpsi1Vf = _dtrtrs(Lm, _dtrtrs(LB, c, trans=1), trans=1)
    dL_dm = -LBi_Lmi_psi1.T @ c + VVT                                        # var_dtc.py:148
    delit = c @ c.T
    data_fit = float(np.trace(delit))
    P = backsub_both_sides(LB, Dy * np.eye(M) + delit)
    dL_dKmm = backsub_both_sides(Lm, -0.5 * P - 0.5 * B * Dy + Dy * np.eye(M))
    dL_dpsi0 = -0.5 * Dy * (beta * np.ones(N))
    dL_dpsi1 = VVT @ Cpsi1Vf.T
    dL_dpsi2_beta = 0.5 * backsub_both_sides(Lm, Dy * np.eye(M) - P)
    dL_dpsi1 = dL_dpsi1 + 2.0 * (psi1 * bcol) @ dL_dpsi2_beta                # :224-226 / :229-233
    if het:                                                                  # :267-269
        lik_1 = -0.5 * N * Dy * np.log(2 * np.pi) + 0.5 * Dy * np.sum(np.log(beta)) - 0.5 * np.sum(beta * np.square(R).sum(-1))
        lik_2 = -0.5 * Dy * (np.sum(beta * psi0) - np.trace(A))
    else:
        lik_1 = -0.5 * N * Dy * (np.log(2.0 * np.pi) - np.log(beta[0])) - 0.5 * beta[0] * float(np.sum(np.square(R)))
        lik_2 = -0.5 * Dy * (np.sum(beta[0] * psi0) - np.trace(A))
    lml = lik_1 + lik_2 - Dy * np.sum(np.log(np.diag(LB))) + 0.5 * data_fit
    if het:                                                                  # :240-256
        LBi = _dtrtrs(LB, np.eye(M))
        Lmi_psi1 = _dtrtrs(Lm, psi1.T)
        b2 = beta[:, None] ** 2
        dL_dR = -0.5 * beta[:, None] + 0.5 * VVT ** 2
        dL_dR += 0.5 * Dy * (psi0 - np.sum(Lmi_psi1 ** 2, 0))[:, None] * b2
        dL_dR += 0.5 * np.sum((LBi.T @ LBi @ Lmi_psi1) * Lmi_psi1, 0)[:, None] * b2
        dL_dR += -(c.T @ LBi_Lmi_psi1).T * R * b2
        dL_dR += 0.5 * (c.T @ LBi_Lmi_psi1).T ** 2 * b2
        dnoise = dL_dR[:, 0] if Dy == 1 else dL_dR
    else:
        b = beta[0]
        trYYT = float(np.sum(np.square(R)))
        dnoise = -0.5 * N * Dy * b + 0.5 * trYYT * b ** 2 + 0.5 * Dy * (psi0.sum() * b ** 2 - np.trace(A) * b)
        dnoise += b * (0.5 * np.sum(A * P) - data_fit)
        dnoise = float(dnoise)
    Bi = -lapack.dpotri(np.asfortranarray(LB), lower=1)[0]
    Bi = np.tril(Bi) + np.tril(Bi, -1).T
    Bi[np.arange(M), np.arange(M)] += 1.0
    woodbury_inv = backsub_both_sides(Lm, Bi)
    # SparseGP._update_gradients (sparse_gp.py:108-118) part by part (add.py:81-84: every part of a sum sees the same dL_dK;
    # prod.py:86-99,101-113: a factor of a product sees it times the other factors' covariance)
    grads, dZ = [None] * len(parts), np.zeros(Z.shape)
    for grp in O._terms(parts):
        for i in grp:
            part = parts[i]
            kind, ARD, var, ls, dims = part[:5]
            Wnm, Wmm, wdiag = dL_dpsi1, dL_dKmm, 1.0
            for j in grp:
                if j != i:
                    Wnm = Wnm * O._part_K(parts[j], X, Z)
                    Wmm = Wmm * O._part_K(parts[j], Z)
                    wdiag *= float(parts[j][2])
            if kind in ("white", "bias"):
                g = float(np.sum(dL_dpsi0)) * wdiag                          # update_gradients_diag (static.py:95-96,172-173)
                if kind == "bias":
                    g += float(np.sum(Wnm)) + float(np.sum(Wmm))             # static.py:169-170
                else:
                    g += float(np.trace(Wmm))                                # White: K(X, Z) = 0, trace for the symmetric call
                grads[i] = np.array([g])
                continue
            Xp, Zp = X[:, dims], Z[:, dims]
            dv = float(np.sum(dL_dpsi0)) * wdiag
            dl = 0.0
            for G_, A_, B_ in ((Wnm, Xp, Zp), (Wmm, Zp, None)):
                a, b_ = O.update_gradients_full(kind, G_, A_, B_, var, ls, ARD)
                dv += float(a)
                dl = dl + np.atleast_1d(np.asarray(b_, float))
            grads[i] = np.concatenate([[dv], dl])
            gz = gradients_X(kind, Wmm, Zp, None, var, ls, ARD) + gradients_X(kind, Wnm.T, Zp, Xp, var, ls, ARD)
            dZ[:, dims] += gz
    return dict(lml=float(lml), dtheta=np.concatenate(grads), dnoise=dnoise, dZ=dZ, woodbury_vector=Cpsi1Vf,
                woodbury_inv=woodbury_inv, dL_dKmm=dL_dKmm, dL_dKnm=dL_dpsi1, dL_dm=dL_dm, Kmm=Kmm, Lm=Lm)


def sparse_predict(parts, Z, Xnew, woodbury_vector, woodbury_inv, full_cov=False):
    """`Posterior._raw_predict` of the sparse posterior (reference posterior.py:220-262): mu = Kx^T wv,
    var = Kxx - Kx^T Winv Kx (diagonal clipped at 1e-15, :248)."""
    Kx = O.sum_kern_K(parts, Z, Xnew)
    mu = Kx.T @ woodbury_vector
    if full_cov:
        return mu, O.sum_kern_K(parts, Xnew) - Kx.T @ (woodbury_inv @ Kx)
    var = O.sum_kdiag(parts) - np.sum((woodbury_inv.T @ Kx) * Kx, 0)
    return mu, np.clip(var, 1e-15, np.inf)[:, None]


def synthetic_Z(X, M, seed=0):
    """Z = X[perm[:M]] (reference models/sparse_gp_regression.py:41-43)"""
    rng = np.random.default_rng(seed + 77)
    return np.ascontiguousarray(X[rng.permutation(X.shape[0])[:M]].copy())

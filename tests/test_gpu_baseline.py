"""GPU (-m gpu): the HIP path against the REFERENCE / the oracle ON THE BASELINE.json CONFIGS THEMSELVES.

    C2  RBF iso N=4096 D=8            full size: reference golden + the oracle run live
    C3  Matern-5/2 ARD D=32           N=4096 live oracle; N=6144 (first size on the overlapped-inverse schedule) and
                                      N=16384 (the headline config) against reference goldens: LML, alpha, ALL 34
                                      gradients, diag(dL_dK), rows of Ky^-1 and L; central differences on 3 lengthscales
    C4  RBF iso N=32768 D=8           single-GPU path and the 2x4 block-cyclic grid (loopback transport) against the
                                      lean-oracle golden
    C5  VarDTC RBF N=200000 M=2048 D=16   reference golden (+ N=20000 and N=70000 against the sparse oracle live)

Fixtures: tests/golden/baseline_*.npz from oracle/make_golden_baseline.py (inputs regenerated from the seed).
Tolerances = the fp64 parity contract of SURVEY.md 8(c): LML rel 1e-10, alpha rel 1e-9, gradients rel 1e-8 of |g|_inf;
sparse: LML rel 1e-9, gradients rel 1e-6.
"""
import numpy as np
import pytest

from conftest import baseline_golden
from gpy_amd import _lib as L
from oracle import gp_oracle as O
from oracle import sparse_oracle as S

pytestmark = pytest.mark.gpu
TOL_LML, TOL_ALPHA, TOL_GRAD = 1e-10, 1e-9, 1e-8


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _check_exact(r, g, fetch=None):
    gref = np.concatenate([g["dvar"], g["dlen"]])
    errs = dict(lml=abs(r["lml"] - g["lml"]) / abs(g["lml"]),
                alpha=np.linalg.norm(r["alpha"] - g["alpha"]) / np.linalg.norm(g["alpha"]),
                grad=np.abs(r["dtheta"] - gref).max() / np.abs(gref).max(),
                dnoise=abs(r["dnoise"] - g["dnoise"][0]) / abs(g["dnoise"][0]),
                diag=np.abs(r["diag_dL_dK"] - g["diag_dL_dK"]).max() / np.abs(g["diag_dL_dK"]).max(),
                logdet=abs(r["logdet"] - g["logdet"]) / abs(g["logdet"]))
    print({k: "%.2e" % v for k, v in errs.items()})
    assert errs["lml"] <= TOL_LML and errs["alpha"] <= TOL_ALPHA and errs["grad"] <= TOL_GRAD
    assert errs["dnoise"] <= TOL_GRAD and errs["diag"] <= TOL_GRAD and errs["logdet"] <= 1e-11
    assert r["dtheta"].size == gref.size
    if fetch is not None:
        rows = g["rows"]
        Lg = fetch(L.FETCH_L)
        assert np.abs(Lg[rows] - g["L_rows"]).max() <= 1e-11
        del Lg
        W = fetch(L.FETCH_KINV)
        assert np.abs(W[rows] - g["Wi_rows"]).max() <= 1e-9 * np.abs(g["Wi_rows"]).max()


def _run_exact(ctx, g):
    X, Y = O.synthetic(g["N"], g["D"], seed=g["seed"])
    th = L.theta_vec(g["variance"], g["lengthscale"], g["ARD"], g["D"])
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference(g["kind"], g["ARD"], th, g["noise"], want_diag=True)
    assert info == 0
    return X, Y, th, r


def test_config2_full_size_against_reference_and_live_oracle(ctx):
    g = baseline_golden("baseline_c2_rbf_n4096_d8")
    X, Y, th, r = _run_exact(ctx, g)
    _check_exact(r, g, fetch=ctx.fetch)
    ref = O.parameters_changed(g["kind"], X, Y, g["variance"], g["lengthscale"], g["ARD"], g["noise"])   # ~7 s of CPU
    assert abs(r["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.linalg.norm(r["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
    assert np.abs(r["dtheta"] - gref).max() <= TOL_GRAD * np.abs(gref).max()
    assert np.abs(ctx.fetch(L.FETCH_KINV) - ref["Wi"]).max() <= 1e-9 * np.abs(ref["Wi"]).max()
    K = L.kern_K(g["kind"], g["ARD"], th, X)
    assert np.abs(K - ref["K"]).max() <= 1e-13 * g["variance"]


def test_config3_kernel_shape_d32_at_n4096_against_live_oracle(ctx):
    """Matern-5/2 ARD with D = 32 exactly (the KDC = 32 staging-chunk boundary of the gradient kernel): all 34 gradients."""
    N, D = 4096, 32
    X, Y = O.synthetic(N, D, seed=1)
    var, ls, noise = O.default_theta(D, True)
    ref = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference("matern52", True, L.theta_vec(var, ls, True, D), noise, want_diag=True)
    assert info == 0
    gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
    assert r["dtheta"].size == 33                 # variance + 32 lengthscales; the 34th gradient is the noise variance
    assert abs(r["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.linalg.norm(r["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    assert np.abs(r["dtheta"] - gref).max() <= TOL_GRAD * np.abs(gref).max()
    assert abs(r["dnoise"] - ref["dL_dnoise"]) <= TOL_GRAD * abs(ref["dL_dnoise"])


@pytest.mark.parametrize("name", ["baseline_c3s_matern52_ard_n6144_d32", "baseline_c3_matern52_ard_n16384_d32"])
def test_config3_against_reference_golden(name, ctx):
    """N=6144: the smallest size on the schedule that inverts the leading block underneath potrf; N=16384: the headline
    configuration (BASELINE configs[2]) itself.  Every one of the 32 ARD lengthscale gradients is compared."""
    g = baseline_golden(name)
    X, Y, th, r = _run_exact(ctx, g)
    _check_exact(r, g, fetch=ctx.fetch)
    if g["N"] < 16384:
        return
    # central differences of the device LML on three lengthscales (first, middle, last) and the noise variance
    for q in (0, 15, 31):
        h = 1e-5 * th[1 + q]
        tp, tm = th.copy(), th.copy()
        tp[1 + q] += h
        tm[1 + q] -= h
        fp = ctx.exact_inference(g["kind"], True, tp, g["noise"], want_alpha=False)[1]["lml"]
        fm = ctx.exact_inference(g["kind"], True, tm, g["noise"], want_alpha=False)[1]["lml"]
        assert abs((fp - fm) / (2 * h) - r["dtheta"][1 + q]) <= 2e-5 * abs(r["dtheta"][1 + q]) + 1e-5
    ctx.set_data(X[:256], Y[:256])


def test_config4_single_gpu_against_lean_oracle_golden(ctx):
    g = baseline_golden("baseline_c4_rbf_n32768_d8")
    X, Y, th, r = _run_exact(ctx, g)
    _check_exact(r, g)                         # no N x N fetch at this size (8.6 GB over PCIe)
    ctx.set_data(X[:256], Y[:256])             # release 26 GB


def test_config4_block_cyclic_grid_2x4_loopback_against_lean_oracle_golden():
    """BASELINE configs[3]'s partitioning (Pr x Pc = 2 x 4, nb = 512) with all eight logical ranks on this one GPU."""
    from gpy_amd import grid as G
    g = baseline_golden("baseline_c4_rbf_n32768_d8")
    X, Y = O.synthetic(g["N"], g["D"], seed=g["seed"])
    gc = G.GridContext.loopback(2, 4, nb=512)
    try:
        gc.set_data(X, Y)
        info, r = gc.exact_inference(g["kind"], g["ARD"], L.theta_vec(g["variance"], g["lengthscale"], False, g["D"]),
                                     g["noise"], want_diag=True)
        assert info == 0
        _check_exact(r, g)
    finally:
        gc.close()


def test_config5_full_size_against_reference_golden():
    g = baseline_golden("baseline_c5_sparse_rbf_n200000_m2048_d16")
    X, Y = O.synthetic(g["N"], g["D"], seed=g["seed"])
    Z = S.synthetic_Z(X, g["M"], g["seed"])
    c = L.SparseContext(0)
    try:
        c.set_data(X, Y)
        info, r = c.vardtc("rbf", False, L.theta_vec(g["variance"], g["lengthscale"], False, g["D"]), Z, g["noise"])
        assert info == 0
        errs = dict(lml=abs(r["lml"] - g["lml"]) / abs(g["lml"]),
                    dtheta=np.abs(r["dtheta"] - g["dtheta"]).max() / np.abs(g["dtheta"]).max(),
                    dnoise=abs(r["dnoise"] - g["dnoise"]) / abs(g["dnoise"]),
                    dZ=np.abs(r["dZ"] - g["dZ"]).max() / np.abs(g["dZ"]).max(),
                    wv=np.linalg.norm(r["woodbury_vector"] - g["woodbury_vector"]) / np.linalg.norm(g["woodbury_vector"]))
        print({k: "%.2e" % v for k, v in errs.items()})
        assert errs["lml"] <= 1e-9 and errs["dtheta"] <= 1e-6 and errs["dnoise"] <= 1e-6 and errs["dZ"] <= 1e-6
        assert errs["wv"] <= 1e-5
    finally:
        c.close()


@pytest.mark.parametrize("N", [20000, 70000])
def test_config5_shape_m2048_d16_against_live_sparse_oracle(N):
    """M = 2048, D = 16 exactly (the k_grad_cols D <= 16 path and the 136-tile split-K Gram) at oracle-affordable N,
    one of them beyond 65536 rows."""
    from test_oracle_sparse import check_sparse
    M, D = 2048, 16
    X, Y = O.synthetic(N, D, seed=4)
    Z = S.synthetic_Z(X, M, 4)
    var, ls, noise = O.default_theta(D, False)
    ref = S.vardtc("rbf", X, Z, Y, var, ls, False, noise)
    c = L.SparseContext(0)
    try:
        c.set_data(X, Y)
        info, r = c.vardtc("rbf", False, L.theta_vec(var, ls, False, D), Z, noise)
        assert info == 0
        check_sparse(r, ref)
    finally:
        c.close()

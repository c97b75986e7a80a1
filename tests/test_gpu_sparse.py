"""GPU (-m gpu): the sparse (VarDTC) path through the C-ABI against the golden vectors generated from the reference's
own VarDTC / SparseGP code and against the CPU oracle on seeded inputs (incl. multi-chunk N and M not a multiple
of 128).  Tolerances (sparse row): LML rel 1e-9, theta / noise / Z gradients rel 1e-6."""
import numpy as np
import pytest

from gpy_amd import _lib as L
from oracle import gp_oracle as O
from oracle import sparse_oracle as S
from test_oracle_sparse import check_sparse, load_sparse_golden, sparse_golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sctx():
    c = L.SparseContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", sparse_golden_names())
def test_sparse_golden_through_the_c_abi(name, sctx):
    g = load_sparse_golden(name)
    ls = g["lengthscale"] if g["ARD"] else g["lengthscale"][:1]
    D = g["X"].shape[1]
    sctx.set_data(g["X"], g["Y"])
    info, r = sctx.vardtc(g["kind"], g["ARD"], L.theta_vec(g["variance"], ls, g["ARD"], D), g["Z"], g["noise"])
    assert info == 0
    check_sparse(r, g)
    dK = sctx.fetch(L.SparseContext.FETCH_DLDKMM)
    assert np.abs(dK - dK.T).max() <= 1e-9 * np.abs(dK).max()
    Wi = sctx.fetch(L.SparseContext.FETCH_WOODBURY_INV)
    assert np.abs(Wi - g["woodbury_inv"]).max() <= 1e-4 * np.abs(g["woodbury_inv"]).max()


@pytest.mark.parametrize("kind,ARD,N,M,D,Dy", [("rbf", True, 300000, 130, 4, 1),      # two chunks (> 262144 rows)
                                               ("matern52", False, 5000, 513, 3, 2)])
def test_sparse_matches_oracle_multichunk(kind, ARD, N, M, D, Dy, sctx):
    X, Y = O.synthetic(N, D, seed=N % 97, Dy=Dy)
    Z = S.synthetic_Z(X, M, 1)
    var, ls, noise = O.default_theta(D, ARD)
    ref = S.vardtc(kind, X, Z, Y, var, ls, ARD, noise)
    sctx.set_data(X, Y)
    info, r = sctx.vardtc(kind, ARD, L.theta_vec(var, ls, ARD, D), Z, noise, want_stage_ms=True)
    assert info == 0
    check_sparse(r, ref)
    P2 = sctx.fetch(L.SparseContext.FETCH_PSI2)
    assert np.abs(P2 - ref["psi2"]).max() <= 1e-11 * np.abs(ref["psi2"]).max()
    # same context, new Z and theta: nothing stale may leak
    Z2 = S.synthetic_Z(X, M, 2)
    ref2 = S.vardtc(kind, X, Z2, Y, 0.8 * var, 1.3 * ls, ARD, 2 * noise)
    info, r2 = sctx.vardtc(kind, ARD, L.theta_vec(0.8 * var, 1.3 * ls, ARD, D), Z2, 2 * noise)
    assert info == 0
    # M = 513 inducing points in D = 3 with longer lengthscales: cond(Kmm + 1e-8 I) ~ 1e12; the reference's own two
    # implementations (VarDTC vs VarDTC_minibatch) only agree to ~1e-4 here -> 1e-5 on the gradients
    check_sparse(r2, ref2, tol_g=1e-5)


def test_gradients_X_through_the_c_abi():
    rng = np.random.default_rng(4)
    for kind, ARD, N, M, D in (("rbf", True, 150, 70, 4), ("matern52", False, 90, 130, 2), ("exponential", True, 64, 64, 3)):
        X, X2 = rng.standard_normal((N, D)), rng.standard_normal((M, D))
        var, ls, _ = O.default_theta(D, ARD)
        th = L.theta_vec(var, ls, ARD, D)
        G = rng.standard_normal((N, M))
        ref = S.gradients_X(kind, G, X, X2, var, ls, ARD)
        got = L.gradients_X(kind, ARD, th, G, X, X2)
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
        Gs = rng.standard_normal((N, N))
        ref = S.gradients_X(kind, Gs, X, None, var, ls, ARD)
        got = L.gradients_X(kind, ARD, th, Gs, X, None)
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


def test_sparse_gp_regression_host_classes_and_checkgrad():
    import gpy_amd
    X, Y = O.synthetic(600, 2, seed=8)
    k = gpy_amd.Matern52(2, variance=1.2, lengthscale=[0.8, 1.4], ARD=True)
    m = gpy_amd.SparseGPRegression(X, Y, kernel=k, num_inducing=24, noise_var=0.15, seed=3)
    ref = S.vardtc("matern52", X, m.Z.values, Y, 1.2, np.array([0.8, 1.4]), True, 0.15)
    assert abs(m.log_likelihood() - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    g = m.gradient
    gref = np.concatenate([ref["dZ"].ravel(), ref["dtheta"], [ref["dnoise"]]])
    assert np.abs(g - gref).max() <= 1e-6 * np.abs(gref).max()
    # finite differences through the whole driver (what GPy's checkgrad does, test_model.py:790-898)
    x0 = m.param_array.copy()
    for i in (0, 5, x0.size - 4, x0.size - 1):
        e = np.zeros_like(x0); e[i] = 1e-6 * max(1.0, abs(x0[i]))
        m.param_array = x0 + e; fp = m.log_likelihood()
        m.param_array = x0 - e; fm = m.log_likelihood()
        m.param_array = x0
        assert abs((fp - fm) / (2 * e[i]) - m.gradient[i]) <= 2e-4 * max(1.0, abs(m.gradient[i]))
    mu, var = m.predict(X[:7])
    mu_ref = O.kern_K("matern52", X[:7], m.Z.values, 1.2, np.array([0.8, 1.4]), True) @ ref["woodbury_vector"]
    assert np.abs(mu - mu_ref).max() <= 1e-6 * np.abs(mu_ref).max()
    assert np.all(var > 0)
    f0 = m.objective_function()
    m.optimize(max_iters=8)
    assert m.objective_function() < f0


def test_row_sharded_mode_rccl_world_one(sctx):
    """The RCCL all-reduce path of the sharded sparse mode on the one GPU we have (world size 1)."""
    from gpy_amd import grid as G
    X, Y = O.synthetic(3000, 4, seed=2)
    Z = S.synthetic_Z(X, 96, 0)
    var, ls, noise = O.default_theta(4, False)
    ref = S.vardtc("rbf", X, Z, Y, var, ls, False, noise)
    c = L.SparseContext(0)
    try:
        c.attach_comm(0, 1, G.unique_id())
        c.set_data(X, Y)
        info, r = c.vardtc("rbf", False, L.theta_vec(var, ls, False, 4), Z, noise)
        assert info == 0
        check_sparse(r, ref)
    finally:
        c.close()

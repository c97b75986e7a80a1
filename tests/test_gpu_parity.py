"""GPU (-m gpu): the HIP path, called through the C-ABI, against (a) the golden vectors generated from the
reference's own code, (b) the CPU oracle on seeded inputs, and (c) size-independent properties at the BASELINE
sizes.  Tolerances are the fp64 parity contract of SURVEY.md 8(c) / BASELINE.md:
    K entries abs <= 1e-13*variance ; LML rel <= 1e-10 ; alpha rel <= 1e-9 ; gradients rel <= 1e-8 (vs |grad|_inf)
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from gpy_amd import _lib as L
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

TOL_K, TOL_LML, TOL_ALPHA, TOL_GRAD = 1e-13, 1e-10, 1e-9, 1e-8


def _ls(g):
    return g["lengthscale"] if g["ARD"] else g["lengthscale"][:1]


def _theta(g):
    return L.theta_vec(g["variance"], _ls(g), g["ARD"], g["X"].shape[1])


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def test_native_library_is_the_one_running():
    assert L.device_count() >= 1
    import ctypes
    assert isinstance(L.lib(), ctypes.CDLL) and L.LIB_PATH.endswith("gpy_amd/libmi355gp.so")
    loaded = open("/proc/self/maps").read()
    assert "libmi355gp.so" in loaded


@pytest.mark.parametrize("name", golden_names())
def test_golden_vectors_through_the_c_abi(name, ctx):
    g = load_golden(name)
    X, Y, th = g["X"], g["Y"], _theta(g)
    N = X.shape[0]
    rows = g["rows"]
    # kernel functions
    K = L.kern_K(g["kind"], g["ARD"], th, X)
    assert np.abs(K[rows] - g["K_rows"]).max() <= TOL_K * g["variance"]
    assert np.array_equal(np.diag(K), np.full(N, g["variance"]))              # exact variance on the diagonal
    Kx = L.kern_K(g["kind"], g["ARD"], th, X, g["X2"])
    assert np.abs(Kx[rows] - g["K_X_X2_rows"]).max() <= TOL_K * g["variance"]
    assert np.array_equal(L.kern_Kdiag(g["kind"], th, N), g["Kdiag"])
    gA = L.update_gradients_full(g["kind"], g["ARD"], th, g["A"], X, g["X2"])
    refA = np.concatenate([g["dvar_A"], g["dlen_A"]])
    assert np.abs(gA - refA).max() <= TOL_GRAD * max(np.abs(refA).max(), 1e-300)
    # fused inference
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference(g["kind"], g["ARD"], th, g["noise"], want_diag=True)
    assert info == 0
    assert abs(r["lml"] - g["lml"]) <= TOL_LML * max(1.0, abs(g["lml"]))
    assert np.linalg.norm(r["alpha"] - g["alpha"]) <= TOL_ALPHA * np.linalg.norm(g["alpha"])
    ref = np.concatenate([g["dvar"], g["dlen"]])
    assert np.abs(r["dtheta"] - ref).max() <= TOL_GRAD * np.abs(ref).max()
    assert np.abs(r["diag_dL_dK"] - g["diag_dL_dK"]).max() <= TOL_GRAD * np.abs(g["diag_dL_dK"]).max()
    if g["noise"].size == 1:
        assert abs(r["dnoise"] - g["dnoise"][0]) <= TOL_GRAD * abs(g["dnoise"][0])
    assert abs(r["logdet"] - g["logdet"]) <= 1e-11 * max(1.0, abs(g["logdet"]))
    # lazily fetched N x N results
    Lg = ctx.fetch(L.FETCH_L, fortran_order=True)
    assert Lg.flags.f_contiguous                                               # what lapack.dpotrf hands GPy
    assert np.abs(Lg[rows] - g["L_rows"]).max() <= 1e-12
    assert np.all(np.triu(Lg, 1) == 0.0)
    G = ctx.fetch(L.FETCH_DLDK)
    assert np.abs(G[rows] - g["dL_dK_rows"]).max() <= 1e-9 * np.abs(g["dL_dK_rows"]).max()
    assert np.array_equal(G, G.T)
    if "pred_mu" in g:
        mu, var = ctx.predict(g["kind"], g["ARD"], th, g["Xs"])
        assert np.abs(mu - g["pred_mu"]).max() <= 1e-10 and np.abs(var - g["pred_var"]).max() <= 1e-10
        _, cov = ctx.predict(g["kind"], g["ARD"], th, g["Xs"], full_cov=True)
        assert np.abs(cov - g["pred_cov"]).max() <= 1e-10


@pytest.mark.parametrize("kind", O.KINDS)
@pytest.mark.parametrize("N,D,Dy,ARD", [(1, 1, 1, False), (2, 3, 1, True), (63, 2, 1, False), (129, 5, 2, True),
                                         (513, 3, 1, True), (1300, 40, 1, True), (2048, 8, 3, False)])
def test_oracle_parity_on_seeded_inputs(kind, N, D, Dy, ARD, ctx):
    X, Y = O.synthetic(N, D, seed=7 * N + D, Dy=Dy)
    var, ls, noise = O.default_theta(D, ARD)
    ref = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference(kind, ARD, L.theta_vec(var, ls, ARD, D), noise, want_diag=True)
    assert info == 0
    assert abs(r["lml"] - ref["lml"]) <= TOL_LML * max(1.0, abs(ref["lml"]))
    assert np.linalg.norm(r["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
    assert np.abs(r["dtheta"] - gref).max() <= TOL_GRAD * np.abs(gref).max()
    assert abs(r["dnoise"] - ref["dL_dnoise"]) <= TOL_GRAD * abs(ref["dL_dnoise"])
    assert np.abs(ctx.fetch(L.FETCH_KINV) - ref["Wi"]).max() <= 1e-9 * np.abs(ref["Wi"]).max()


def test_heteroscedastic_noise_and_given_K(ctx):
    N, D = 400, 3
    X, Y = O.synthetic(N, D, seed=3, Dy=2)
    nv = 0.05 + 0.1 * np.random.default_rng(0).random(N)
    K = O.kern_K("matern32", X, None, 1.3, 0.9, False)
    ref = O.exact_inference(K, Y, nv)
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference("matern32", False, np.array([1.3, 0.9]), nv, want_diag=True)
    assert info == 0 and abs(r["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.abs(r["diag_dL_dK"] - ref["diag_dL_dK"]).max() <= TOL_GRAD * np.abs(ref["diag_dL_dK"]).max()
    # the K= argument of ExactGaussianInference.inference (reference test_inference.py:118-133)
    info, r2 = ctx.inference_given_K(K, nv, want_diag=True)
    assert info == 0 and abs(r2["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.linalg.norm(r2["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])


def test_potrf_pdinv_and_info_codes():
    import scipy.linalg as sla
    for n in (5, 128, 200, 777):
        X, _ = O.synthetic(n, 4, seed=n)
        A = O.kern_K("rbf", X, None, 1.0, 1.5, False) + 0.05 * np.eye(n)
        Lg, info, _ = L.potrf(A)
        assert info == 0 and np.abs(Lg - sla.cholesky(A, lower=True)).max() <= 1e-12
        Ai, L2, logdet, info, _ = L.pdinv(A)
        Air, Lr, _, ldr = O.pdinv(A)
        assert info == 0 and np.abs(Ai - Air).max() <= 1e-9 * np.abs(Air).max() and abs(logdet - ldr) <= 1e-10 * max(1, abs(ldr))
        assert np.array_equal(Ai, Ai.T)
    A = np.eye(300)
    A[140, 140] = -2.0
    assert L.potrf(A)[1] == 141                               # LAPACK-style info: first failing leading minor
    A = np.ones((260, 260))                                    # rank one: fails at the second pivot
    assert L.potrf(A)[1] == 2


def test_jitter_ladder_and_linalg_errors_through_the_host_classes():
    import gpy_amd
    X, Y = O.synthetic(150, 2, seed=5)
    Xd, Yd = np.vstack([X, X]), np.vstack([Y, Y])             # duplicated rows + (almost) no noise: singular Ky
    m = gpy_amd.GPRegression(Xd, Yd, gpy_amd.RBF(2, variance=1.0, lengthscale=1.0), noise_var=1e-300)
    assert np.isfinite(m.log_likelihood())
    ref = O.exact_inference(O.kern_K("rbf", Xd, None, 1.0, 1.0, False), Yd, 1e-300)   # the oracle runs the same ladder
    assert abs(m.log_likelihood() - ref["lml"]) <= 1e-3 * abs(ref["lml"])             # ill-conditioned by construction
    inf = gpy_amd.ExactGaussianInference()
    Kbad = -np.eye(6)
    with pytest.raises(np.linalg.LinAlgError, match="non-positive diagonal"):
        inf.inference(None, X[:6], gpy_amd.Gaussian(0.1), Y[:6], K=Kbad)
    Kind = np.array([[1.0, 5.0], [5.0, 1.0]])
    with pytest.raises(np.linalg.LinAlgError, match="even with jitter"):
        inf.inference(None, X[:2], gpy_amd.Gaussian(1e-12), Y[:2], K=Kind)


def test_drop_in_classes_match_oracle_and_finite_differences():
    import gpy_amd
    N, D = 600, 3
    X, Y = O.synthetic(N, D, seed=9, Dy=2)
    var, ls, noise = 0.8, np.array([0.6, 1.1, 1.9]), 0.15
    for cls, kind in ((gpy_amd.RBF, "rbf"), (gpy_amd.Matern52, "matern52"), (gpy_amd.Matern32, "matern32"),
                      (gpy_amd.Exponential, "exponential")):
        m = gpy_amd.GPRegression(X, Y, cls(D, variance=var, lengthscale=ls, ARD=True), noise_var=noise)
        ref = O.parameters_changed(kind, X, Y, var, ls, True, noise)
        gref = np.concatenate([[ref["dvar"]], ref["dlen"], [ref["dL_dnoise"]]])
        assert abs(m.log_likelihood() - ref["lml"]) <= TOL_LML * abs(ref["lml"])
        assert np.abs(m.gradient - gref).max() <= TOL_GRAD * np.abs(gref).max()
        # the style of the reference's own checks: central differences of the LML (checkgrad)
        x0 = m.param_array.copy()
        g0 = m.gradient.copy()
        fd = np.zeros_like(x0)
        for i in range(x0.size):
            h = 1e-6 * x0[i]
            xp, xm = x0.copy(), x0.copy()
            xp[i] += h
            xm[i] -= h
            m.param_array = xp
            fp = m.log_likelihood()
            m.param_array = xm
            fm = m.log_likelihood()
            fd[i] = (fp - fm) / (2 * h)
        m.param_array = x0
        assert np.allclose(g0, fd, rtol=5e-5, atol=1e-5)
        # a foreign consumer sees plain arrays: the same gradients through the generic (host dL_dK) entry point
        k2 = cls(D, variance=var, lengthscale=ls, ARD=True)
        k2.update_gradients_full(np.asarray(m.grad_dict["dL_dK"]), X)
        assert np.abs(np.concatenate([k2.variance.gradient, k2.lengthscale.gradient]) - gref[:-1]).max() <= TOL_GRAD * np.abs(gref).max()
        # posterior members (reference posterior.py) and predictions (gp.py:308-365)
        assert np.abs(np.asarray(m.posterior.woodbury_chol) - ref["L"]).max() <= 1e-11
        Xs = np.random.default_rng(1).standard_normal((50, D))
        mu, v = m.predict(Xs)
        mur, vr = O.predict(kind, X, Xs, ref["L"], ref["alpha"], var, ls, True, noise=noise)
        assert np.abs(mu - mur).max() <= 1e-9 and np.abs(v - vr).max() <= 1e-9


def test_inv_lengthscale_and_active_dims_through_device():
    import gpy_amd
    X, Y = O.synthetic(300, 4, seed=2)
    k = gpy_amd.RBF(2, variance=1.2, lengthscale=0.8, inv_l=True, active_dims=[1, 3])
    m = gpy_amd.GPRegression(X, Y, k, noise_var=0.2)
    ref = O.parameters_changed("rbf", np.ascontiguousarray(X[:, [1, 3]]), Y, 1.2, 0.8, False, 0.2)
    assert abs(m.log_likelihood() - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.allclose(k.inv_l.gradient, ref["dlen"] * (0.8 ** 3 / -2.0), rtol=1e-8)      # reference rbf.py:373-375


def test_results_are_bit_reproducible_and_independent_of_lookahead(ctx):
    X, Y = O.synthetic(3000, 6, seed=4)
    var, ls, noise = O.default_theta(6, True)
    th = L.theta_vec(var, ls, True, 6)
    ctx.set_data(X, Y)
    outs = []
    ctx.set_option("persist", 1)                 # the factorisation alone as one launch: the bits of the launch-per-step schedules
    for la in (1, 1, 0):
        ctx.set_option("lookahead", la)
        info, r = ctx.exact_inference("matern52", True, th, noise)
        outs.append((r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes(), r["dnoise"]))
    ctx.set_option("lookahead", 1)
    assert outs[0] == outs[1] == outs[2]
    ctx.set_option("persist", -1)


@pytest.mark.parametrize("opts", [{"tri_overlap": 0}, {"tri_min_nt": 16, "tri_h": 8},
                                  {"tri_min_nt": 16, "tri_h": 8, "tri_half": 0},
                                  {"nbo": 256}, {"diag_excl_first": 0, "solve_overlap": 0}])
def test_schedule_switches_give_the_same_factorisation(opts):
    """The schedule switches that remain (per context through `mi355gp_set_option` -- the product library reads
    none of them from the environment) change how the work is launched -- the overlapped leading inverse, its side stream,
    the outer panel width -- not what is computed: same LML / alpha / gradients as the default to rounding."""
    X, Y = O.synthetic(2900, 5, seed=7)
    var, ls, noise = O.default_theta(5, True)
    th = L.theta_vec(var, ls, True, 5)
    c0 = L.Context(0)
    try:
        c0.set_data(X, Y)
        _, ref = c0.exact_inference("rbf", True, th, noise)
    finally:
        c0.close()
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        for k, v in opts.items():
            c.set_option(k, v)
            assert c.get_option(k) == v
        for _ in range(2):                                   # second call: flags carry a new generation
            info, r = c.exact_inference("rbf", True, th, noise)
            assert info == 0
            assert abs(r["lml"] - ref["lml"]) <= 1e-12 * abs(ref["lml"])
            assert np.abs(r["alpha"] - ref["alpha"]).max() <= 1e-11 * np.abs(ref["alpha"]).max()
            assert np.abs(r["dtheta"] - ref["dtheta"]).max() <= 1e-10 * np.abs(ref["dtheta"]).max()
        c.set_option("lookahead", 0)                         # serial schedule: must not depend on the server / flags
        info, r = c.exact_inference("rbf", True, th, noise)
        assert info == 0 and abs(r["lml"] - ref["lml"]) <= 1e-12 * abs(ref["lml"])
    finally:
        c.close()


@pytest.mark.parametrize("n", [6200])
def test_pdinv_with_overlapped_leading_inverse_at_ragged_sizes(n):
    """N >= 6144 takes the schedule that inverts the leading block underneath potrf's second half; tile
    count that is not a power of two (49) exercises the sub-range merge levels.  Size-independent checks:
    A Ainv v = v, L L^T v = A v, log det against LAPACK."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 4))
    s = (X * X).sum(1)
    A = np.exp(-0.125 * np.maximum(s[:, None] + s[None, :] - 2.0 * X @ X.T, 0.0)) + 0.1 * np.eye(n)
    Ai, Lc, logdet, info, _ = L.pdinv(A)
    assert info == 0
    V = rng.standard_normal((n, 3))
    assert np.abs(A @ (Ai @ V) - V).max() <= 1e-9
    assert np.abs(Lc @ (Lc.T @ V) - A @ V).max() <= 1e-12 * np.abs(A @ V).max()
    assert abs(np.linalg.slogdet(A)[1] - logdet) <= 1e-9 * abs(logdet)
    assert np.abs(Ai - Ai.T).max() == 0.0


@pytest.mark.parametrize("kind,ARD,N,D", [("rbf", False, 4096, 8), ("matern52", True, 16384, 32)])
def test_size_independent_properties_at_baseline_sizes(kind, ARD, N, D, ctx):
    """BASELINE configs[1] and configs[2]: properties that need no O(N^3) CPU reference."""
    X, Y = O.synthetic(N, D, seed=0)
    var, ls, noise = O.default_theta(D, ARD)
    th = L.theta_vec(var, ls, ARD, D)
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference(kind, ARD, th, noise, want_diag=True)
    assert info == 0
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(N, 24, replace=False))
    Ky_rows = L.kern_K(kind, ARD, th, X[rows], X)
    Ky_rows[np.arange(rows.size), rows] += noise + 1e-8
    # (1) Ky alpha = Y on sampled rows
    assert np.abs(Ky_rows @ r["alpha"] - Y[rows]).max() <= 1e-9 * max(1.0, np.abs(Y).max())
    # (2) L L^T = Ky and Ky^-1 Ky = I on sampled rows
    Lg = ctx.fetch(L.FETCH_L)
    assert np.abs(Lg[rows] @ Lg.T - Ky_rows).max() <= 1e-11 * var
    assert abs(2.0 * np.sum(np.log(np.diag(Lg))) - r["logdet"]) <= 1e-9 * abs(r["logdet"])
    del Lg
    W = ctx.fetch(L.FETCH_KINV)
    Ky_cols = Ky_rows.T                                        # Ky is symmetric: columns `rows` of Ky
    E = W @ Ky_cols
    E[rows, np.arange(rows.size)] -= 1.0
    assert np.abs(E).max() <= 1e-8
    assert abs(np.trace(W) - r["trKinv"]) <= 1e-9 * abs(r["trKinv"])
    # (3) dnoise = trace(dL_dK) = 0.5(|alpha|^2 - tr Ky^-1);  LML from its parts
    assert abs(r["dnoise"] - 0.5 * (np.sum(r["alpha"] ** 2) - np.trace(W))) <= 1e-9 * abs(r["dnoise"])
    assert abs(r["diag_dL_dK"].sum() - r["dnoise"]) <= 1e-9 * abs(r["dnoise"])
    lml = 0.5 * (-N * np.log(2 * np.pi) - r["logdet"] - float(np.sum(r["alpha"] * Y)))
    assert abs(lml - r["lml"]) <= 1e-12 * abs(lml)
    del W
    # (4) the variance gradient against a central difference of the device LML (one parameter: two more evaluations)
    h = 1e-5 * var
    fp = ctx.exact_inference(kind, ARD, L.theta_vec(var + h, ls, ARD, D), noise)[1]["lml"]
    fm = ctx.exact_inference(kind, ARD, L.theta_vec(var - h, ls, ARD, D), noise)[1]["lml"]
    assert abs((fp - fm) / (2 * h) - r["dtheta"][0]) <= 1e-5 * abs(r["dtheta"][0]) + 1e-6


def test_config4_size_on_one_gpu_without_fetching_n_squared(ctx):
    """BASELINE configs[3] size (RBF, N=32768, D=8) on a single MI355X (3 x 8.6 GB resident): the checks that need
    no N^2 transfer -- Ky alpha = Y on sampled rows, LML from its parts, variance gradient by central differences."""
    kind, ARD, N, D = "rbf", False, 32768, 8
    X, Y = O.synthetic(N, D, seed=0)
    var, ls, noise = O.default_theta(D, ARD)
    th = L.theta_vec(var, ls, ARD, D)
    ctx.set_data(X, Y)
    info, r = ctx.exact_inference(kind, ARD, th, noise, want_diag=True)
    assert info == 0
    rows = np.sort(np.random.default_rng(2).choice(N, 16, replace=False))
    Ky_rows = L.kern_K(kind, ARD, th, X[rows], X)
    Ky_rows[np.arange(rows.size), rows] += noise + 1e-8
    assert np.abs(Ky_rows @ r["alpha"] - Y[rows]).max() <= 1e-9 * max(1.0, np.abs(Y).max())
    lml = 0.5 * (-N * np.log(2 * np.pi) - r["logdet"] - float(np.sum(r["alpha"] * Y)))
    assert abs(lml - r["lml"]) <= 1e-12 * abs(lml)
    assert abs(r["diag_dL_dK"].sum() - r["dnoise"]) <= 1e-9 * abs(r["dnoise"])
    assert abs(r["dnoise"] - 0.5 * (np.sum(r["alpha"] ** 2) - r["trKinv"])) <= 1e-9 * abs(r["dnoise"])
    h = 1e-5 * var
    fp = ctx.exact_inference(kind, ARD, L.theta_vec(var + h, ls, ARD, D), noise)[1]["lml"]
    fm = ctx.exact_inference(kind, ARD, L.theta_vec(var - h, ls, ARD, D), noise)[1]["lml"]
    assert abs((fp - fm) / (2 * h) - r["dtheta"][0]) <= 1e-5 * abs(r["dtheta"][0]) + 1e-6
    ctx.set_data(X[:256], Y[:256])            # release the 26 GB before the next test


def test_loo_matches_reference_formula_without_fetching_the_inverse():
    """ExactGaussianInference.LOO (reference exact_gaussian_inference.py:76-88) from the device's diag(dL_dK)."""
    import gpy_amd
    X, Y = O.synthetic(400, 3, seed=6)
    k = gpy_amd.RBF(3, variance=1.2, lengthscale=0.9)
    m = gpy_amd.GPRegression(X, Y, k, noise_var=0.1)
    loo = m.inference_method.LOO(k, X, Y, m.likelihood, m.posterior)
    ref = O.parameters_changed("rbf", X, Y, 1.2, 0.9, False, 0.1)
    c = np.diag(ref["Wi"])[:, None]
    expect = -(0.5 * np.log(2 * np.pi) - 0.5 * np.log(c) + 0.5 * ref["alpha"] ** 2 / c)
    assert loo.shape == expect.shape and np.abs(loo - expect).max() <= 1e-9 * np.abs(expect).max()


def test_covariance_between_points_on_device():
    """Posterior.covariance_between_points (reference posterior.py:109-130)."""
    import gpy_amd
    from scipy.linalg import solve_triangular
    X, Y = O.synthetic(500, 4, seed=7)
    var, ls, noise = O.default_theta(4, True)
    k = gpy_amd.Matern52(4, variance=var, lengthscale=ls, ARD=True)
    m = gpy_amd.GPRegression(X, Y, k, noise_var=noise)
    rng = np.random.default_rng(3)
    X1, X2 = rng.standard_normal((37, 4)), rng.standard_normal((150, 4))
    got = m.posterior.covariance_between_points(k, X, X1, X2)
    ref = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
    t1 = solve_triangular(ref["L"], O.kern_K("matern52", X, X1, var, ls, True), lower=True)
    t2 = solve_triangular(ref["L"], O.kern_K("matern52", X, X2, var, ls, True), lower=True)
    expect = O.kern_K("matern52", X1, X2, var, ls, True) - t1.T @ t2
    assert got.shape == (37, 150) and np.abs(got - expect).max() <= 1e-10


def test_bench_factor_entry_point_reports_sane_rates():
    r = L.bench_factor(2048, reps=2)
    assert all(r[k] > 0 for k in ("potrf_ms", "trtri_ms", "lauum_ms"))
    assert 0.5 < r["lauum_tflops"] < 78.6 and 0.5 < r["potrf_tflops"] < 78.6


def test_integration_stub_against_the_real_library():
    """integration/gpy_mi355x.py (the reference-side ctypes binding of INTEGRATION.md) bound to the real libmi355gp.so, with
    gpy_amd's paramz-free classes standing in for GPy's (the CPU suite runs the same file against the reference's own
    classes with the library mocked by the oracle: tests/test_integration_stub.py)."""
    import os
    import sys
    import types
    import gpy_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration"))
    import gpy_mi355x
    ns = types.SimpleNamespace(RBF=gpy_amd.RBF, Matern52=gpy_amd.Matern52, Matern32=gpy_amd.Matern32,
                               Exponential=gpy_amd.Exponential, PosteriorExact=gpy_amd.PosteriorExact,
                               LatentFunctionInference=object)
    C = gpy_mi355x.make_classes(gpy_mi355x.bind(L.LIB_PATH), ns)
    X, Y = O.synthetic(700, 4, seed=12, Dy=2)
    var, ls, noise = O.default_theta(4, True)
    ref = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
    k = C.Matern52(4, variance=var, lengthscale=ls, ARD=True)
    lik = gpy_amd.Gaussian(variance=noise)
    post, lml, gd = C.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    gref = np.concatenate([[ref["dvar"]], ref["dlen"], [ref["dL_dnoise"]]])
    got = np.concatenate([k.variance.gradient, k.lengthscale.gradient, lik.variance.gradient])
    assert abs(lml - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.linalg.norm(post.woodbury_vector - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    assert np.abs(got - gref).max() <= TOL_GRAD * np.abs(gref).max()
    assert np.abs(np.asarray(post.woodbury_chol) - ref["L"]).max() <= 1e-11
    assert np.abs(np.asarray(gd["dL_dK"]) - ref["dL_dK"]).max() <= 1e-9 * np.abs(ref["dL_dK"]).max()
    assert np.abs(k.K(X) - ref["K"]).max() <= TOL_K * var


class _LinearMean(object):                                       # a mean function in GPy's shape: f(X), update_gradients
    def __init__(self, w):
        self.w, self.grad = w, None

    def f(self, X):
        return X @ self.w

    def update_gradients(self, dL_dm, X):
        self.grad = X.T @ dL_dm


def test_mean_function_Z_tilde_set_targets_and_pickling():
    """The branches of ExactGaussianInference.inference that the golden cases do not reach: `mean_function`
    (exact_gaussian_inference.py:42-50, core/gp.py:281-282), `Z_tilde` (:64-68), new targets on unchanged inputs
    (mi355gp_set_targets), and pickling after an inference call (device handles are dropped, lazy results materialise)."""
    import pickle
    import gpy_amd
    Linear = _LinearMean
    X, Y = O.synthetic(500, 3, seed=13)
    w = np.array([[0.3], [-0.2], [0.1]])
    var, ls, noise = O.default_theta(3, False)
    K = O.kern_K("rbf", X, None, var, ls, False)
    ref = O.exact_inference(K, Y, noise, mean=X @ w, Z_tilde=-2.5)
    mf = Linear(w)
    k = gpy_amd.RBF(3, variance=var, lengthscale=ls)
    inf = gpy_amd.ExactGaussianInference()
    post, lml, gd = inf.inference(k, X, gpy_amd.Gaussian(noise), Y, mean_function=mf, Z_tilde=-2.5)
    assert abs(lml - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.linalg.norm(gd["dL_dm"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    m = gpy_amd.GP(X, Y, k, gpy_amd.Gaussian(noise), mean_function=mf)
    assert np.abs(mf.grad - X.T @ ref["alpha"]).max() <= 1e-8 * np.abs(X.T @ ref["alpha"]).max()
    mu, _ = m.predict(X[:5])
    assert np.abs(mu - (K[:5] @ ref["alpha"] + X[:5] @ w)).max() <= 1e-9
    # same X, new Y: only the targets are re-uploaded (mi355gp_set_targets)
    st = inf._state
    Y2 = Y + 0.5
    ref2 = O.exact_inference(K, Y2, noise)
    _, lml2, _ = inf.inference(k, X, gpy_amd.Gaussian(noise), Y2)
    assert inf._state is st and abs(lml2 - ref2["lml"]) <= TOL_LML * abs(ref2["lml"])
    # pickling after a call: the inference object, the posterior and a lazy N x N result
    inf2 = pickle.loads(pickle.dumps(inf))
    assert inf2._state is None and inf2._last is None
    post2, lml3, gd3 = inf.inference(k, X, gpy_amd.Gaussian(noise), Y2)
    postp = pickle.loads(pickle.dumps(post2))
    assert postp._state is None and np.abs(np.asarray(postp.woodbury_chol) - ref2["L"]).max() <= 1e-11
    Gp = pickle.loads(pickle.dumps(gd3["dL_dK"]))
    assert np.abs(np.asarray(Gp) - ref2["dL_dK"]).max() <= 1e-9 * np.abs(ref2["dL_dK"]).max()
    mp = pickle.loads(pickle.dumps(m))
    assert abs(mp.log_likelihood() - m.log_likelihood()) == 0.0
    _, lml4, _ = inf2.inference(k, X, gpy_amd.Gaussian(noise), Y2)           # the unpickled object builds a fresh context
    assert abs(lml4 - ref2["lml"]) <= TOL_LML * abs(ref2["lml"])


def test_expquad_kernel_cache_and_heteroscedastic_likelihood():
    import gpy_amd
    X, Y = O.synthetic(300, 2, seed=14)
    k = gpy_amd.ExpQuad(2, variance=1.1, lengthscale=0.9)                    # stationary.py:623-662: RBF's function
    assert k.to_dict()["class"] == "GPy.kern.ExpQuad"
    K1 = k.K(X)
    assert np.abs(K1 - O.kern_K("rbf", X, None, 1.1, 0.9, False)).max() <= TOL_K * 1.1
    assert k.K(X) is K1                                                      # Cache_this(limit=3) on K (stationary.py:105)
    k.lengthscale[:] = 1.0
    K2 = k.K(X)
    assert K2 is not K1 and np.abs(K2 - O.kern_K("rbf", X, None, 1.1, 1.0, False)).max() <= TOL_K * 1.1
    # per-point noise through the likelihood class (likelihoods/gaussian.py:347-362)
    meta = {"output_index": np.arange(300)[:, None]}
    lik = gpy_amd.HeteroscedasticGaussian(meta, variance=0.1)
    lik.variance[:] = 0.05 + 0.1 * np.random.default_rng(1).random(300)
    m = gpy_amd.GP(X, Y, k, lik, Y_metadata=meta)
    ref = O.exact_inference(O.kern_K("rbf", X, None, 1.1, 1.0, False), Y, lik.variance.values)
    assert abs(m.log_likelihood() - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    assert np.abs(lik.variance.gradient - ref["diag_dL_dK"]).max() <= TOL_GRAD * np.abs(ref["diag_dL_dK"]).max()


def test_graph_replay_is_bit_identical_to_plain_launches_and_survives_set_data():
    """Below the overlapped-inverse threshold the factorisation region of a context's 2nd evaluation is captured into a hipGraph
    and replayed from the 3rd on: every replay, with theta / noise changing between calls, must give the
    bits of a fresh context's first (plain) evaluation; set_data drops the graph (its nodes point into the old buffers)."""
    rng = np.random.default_rng(12)
    X, Y = O.synthetic(1800, 4, seed=3)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        for it in range(6):
            var, noise = 0.8 + 0.3 * it, 0.05 + 0.02 * it
            ls = rng.uniform(0.6, 2.0, 4)
            th = L.theta_vec(var, ls, True, 4)
            info, r = c.exact_inference("matern32", True, th, noise)
            f = L.Context(0)
            try:
                f.set_data(X, Y)
                info2, r2 = f.exact_inference("matern32", True, th, noise)          # first call of a context: plain launches
            finally:
                f.close()
            assert info == info2 == 0
            assert r["lml"] == r2["lml"] and r["dtheta"].tobytes() == r2["dtheta"].tobytes()
            assert r["alpha"].tobytes() == r2["alpha"].tobytes() and r["dnoise"] == r2["dnoise"]
        # a call with stage timings runs plain and leaves the graph in place; the next one replays again
        info, r3 = c.exact_inference("matern32", True, th, noise, want_stage_ms=True)
        info, r4 = c.exact_inference("matern32", True, th, noise)
        assert r3["lml"] == r4["lml"] == r["lml"] and r3["stage_ms"]["potrf"] > 0
        # new data of another size: the old graph must be gone
        X2, Y2 = O.synthetic(1300, 4, seed=4)
        c.set_data(X2, Y2)
        for _ in range(3):
            info, r5 = c.exact_inference("matern32", True, th, noise)
        ref = O.parameters_changed("matern32", X2, Y2, var, ls, True, noise)
        assert abs(r5["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
        assert np.linalg.norm(r5["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    finally:
        c.close()


def test_schedule_switches_through_the_c_abi_per_context():
    """a context's schedule is configured through `mi355gp_set_option` (not the environment), the value
    survives `set_data`, -1 returns to the process default, and two contexts of one process hold different settings."""
    X, Y = O.synthetic(2900, 5, seed=7)
    var, ls, noise = O.default_theta(5, True)
    th = L.theta_vec(var, ls, True, 5)
    a, b = L.Context(0), L.Context(0)
    try:
        a.set_data(X, Y)
        b.set_data(X, Y)
        _, ref = a.exact_inference("rbf", True, th, noise)
        dflt = {k: b.get_option(k) for k in ("nbo", "tri_min_nt", "tri_h", "tri_overlap", "graph", "solve_overlap", "persist")}
        assert dflt["nbo"] == 0 and dflt["tri_overlap"] == 1 and dflt["graph"] == 1
        for opts in ({"nbo": 256}, {"tri_min_nt": 16, "tri_h": 8}, {"tri_min_nt": 16, "tri_h": 8, "tri_half": 0},
                     {"tri_overlap": 0, "graph": 0}, {"diag_excl_first": 0, "solve_overlap": 0}):
            for k, v in opts.items():
                b.set_option(k, v)
                assert b.get_option(k) == v
            b.set_data(X, Y)                                         # a new workspace: the options must still hold
            for k, v in opts.items():
                assert b.get_option(k) == v
            assert a.get_option("nbo") == 0 and a.get_option("tri_h") == dflt["tri_h"]      # the other context is untouched
            for _ in range(3):                                       # plain, capture, replay where the graph applies
                info, r = b.exact_inference("rbf", True, th, noise)
                assert info == 0
                assert abs(r["lml"] - ref["lml"]) <= 1e-12 * abs(ref["lml"])
                assert np.abs(r["alpha"] - ref["alpha"]).max() <= 1e-11 * np.abs(ref["alpha"]).max()
                assert np.abs(r["dtheta"] - ref["dtheta"]).max() <= 1e-10 * np.abs(ref["dtheta"]).max()
            for k in opts:
                b.set_option(k, -1)
                assert b.get_option(k) == dflt.get(k, b.get_option(k))
        with pytest.raises(L.MI355GPError):
            b.set_option("nbo", 200)                                 # not a multiple of 128
    finally:
        a.close()
        b.close()


def test_in_place_edits_of_caller_buffers_reach_the_device():
    """`X_buf[i] = x_new; m.set_XY(X_buf, Y_buf)` and an in-place edit of an array handed straight to
    `inference()` must never leave stale data on the device (the sampled identity token of round 2 missed them)."""
    import gpy_amd
    X, Y = O.synthetic(1500, 3, seed=21)
    var, ls, noise = O.default_theta(3, False)
    Xb, Yb = X.copy(), Y.copy()
    m = gpy_amd.GPRegression(Xb, Yb, gpy_amd.RBF(3, variance=var, lengthscale=float(ls[0])), noise_var=noise)
    assert not m.X.flags.writeable and m.X is not Xb                 # the model holds frozen private copies
    with pytest.raises(ValueError):
        m.X[0, 0] = 1.0
    Xb[777, 1] += 0.25                                               # an element no 64-sample fingerprint of 4500 looks at
    Yb[1234, 0] -= 0.5
    m.set_XY(Xb, Yb)
    ref = O.parameters_changed("rbf", Xb, Yb, var, ls, False, noise)
    assert abs(m.log_likelihood() - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    # the inference method called directly with WRITABLE arrays (what GPy's own GP class would do with plain ndarrays)
    inf = gpy_amd.ExactGaussianInference()
    k, lik = gpy_amd.RBF(3, variance=var, lengthscale=float(ls[0])), gpy_amd.Gaussian(variance=noise)
    _, l1, _ = inf.inference(k, Xb, lik, Yb)
    assert abs(l1 - ref["lml"]) <= TOL_LML * abs(ref["lml"])
    Xb[3, 2] -= 0.125
    _, l2, _ = inf.inference(k, Xb, lik, Yb)                         # same object, same address, one element changed
    ref2 = O.parameters_changed("rbf", Xb, Yb, var, ls, False, noise)
    assert abs(l2 - ref2["lml"]) <= TOL_LML * abs(ref2["lml"]) and l2 != l1
    Yb[5, 0] += 1.0
    _, l3, _ = inf.inference(k, Xb, lik, Yb)
    ref3 = O.parameters_changed("rbf", Xb, Yb, var, ls, False, noise)
    assert abs(l3 - ref3["lml"]) <= TOL_LML * abs(ref3["lml"])
    # kernel-matrix cache: an edited buffer must not hit
    K1 = k.K(Xb)
    Xb[100, 0] += 0.5
    K2 = k.K(Xb)
    assert K2 is not K1 and np.abs(K2 - O.kern_K("rbf", Xb, None, var, ls, False)).max() <= TOL_K * var


@pytest.mark.parametrize("N", [3100, 3600])
def test_inverse_underneath_the_persistent_launch_with_a_ragged_trailing_block(N):
    """Round 6: the block inverted on the side stream underneath the persistent Cholesky is the largest power of two <= 2 nt / 3
    tiles (nt = 25: 16 of 25, nt = 29: 16 of 29), so the trailing block is SHORTER than the leading one and the top-level pair is
    ragged (`trtri_rows_last`).  The early inverse starts with a context's second evaluation: evaluations 2 .. 4 must agree with the
    oracle to the parity contract and with each other bit for bit."""
    D = 4
    X, Y = O.synthetic(N, D, seed=N)
    var, ls, noise = O.default_theta(D, True)
    th = L.theta_vec(var, ls, True, D)
    ref = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
    gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        outs = []
        for _ in range(4):
            info, r = c.exact_inference("matern52", True, th, noise)
            assert info == 0
            outs.append(r)
        for r in outs[1:]:
            assert abs(r["lml"] - ref["lml"]) <= TOL_LML * abs(ref["lml"])
            assert np.linalg.norm(r["alpha"].ravel() - ref["alpha"].ravel()) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
            assert np.abs(r["dtheta"] - gref).max() <= TOL_GRAD * np.abs(gref).max()
            assert r["lml"] == outs[1]["lml"] and r["dtheta"].tobytes() == outs[1]["dtheta"].tobytes()
        assert c.get_option("persist_aborts") == 0
    finally:
        c.close()

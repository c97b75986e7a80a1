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


def test_more_than_32_input_dimensions_in_gradients_X_and_the_sparse_path(sctx):
    """`stationary.py:245-252,330-358` has no limit on the input dimension; round 2 had D <= 32 in
    `mi355gp_gradients_X` and the sparse path (the column reductions H^T [x~ | 1] now run in groups of 32 dimensions).
    D = 40 (two groups, a ragged second one), D = 64 (the all-ones column in a launch of its own) and D = 33."""
    rng = np.random.default_rng(11)
    for kind, ARD, N, M, D in (("rbf", True, 120, 90, 40), ("matern32", True, 70, 100, 64), ("matern52", False, 80, 60, 33)):
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
    for kind, ARD, N, M, D, Dy in (("rbf", True, 3000, 70, 40, 1), ("matern52", True, 1500, 45, 64, 2)):
        X, Y = O.synthetic(N, D, seed=D, Dy=Dy)
        Z = S.synthetic_Z(X, M, 3)
        var, ls, noise = O.default_theta(D, ARD)
        ref = S.vardtc(kind, X, Z, Y, var, ls, ARD, noise)
        sctx.set_data(X, Y)
        info, r = sctx.vardtc(kind, ARD, L.theta_vec(var, ls, ARD, D), Z, noise)
        assert info == 0
        check_sparse(r, ref)


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


# ---- sum kernels, per-point noise, mean function, device prediction, dL_dKnm blocks --------------------------
from test_oracle_sparse import check_sparse2, load_sparse2_golden, sparse2_golden_names  # noqa: E402


def _specs_of(g):
    return [(p[0], p[1], L.theta_vec(p[2], p[3], p[1], len(p[4])) if p[3] is not None else np.array([p[2]]),
             np.asarray(p[4], np.int32), p[5] if len(p) > 5 else 0) for p in g["parts"]]


@pytest.mark.parametrize("name", sparse2_golden_names())
def test_sparse_sum_hetero_meanfn_golden_through_the_c_abi(name, sctx):
    """`mi355gp_vardtc_inference_sum` / `mi355gp_sparse_predict` / `mi355gp_sparse_fetch_dLdKnm` against outputs of the
    reference's own VarDTC + SparseGP._update_gradients + Posterior._raw_predict (oracle/make_golden_sparse2.py; round 3,
    oracle/make_golden_sparse3.py: per-point noise with several output columns, D = 40, product kernels)."""
    g = load_sparse2_golden(name)
    specs = _specs_of(g)
    sctx.set_data(g["X"], g["R"])
    info, r = sctx.vardtc_sum(specs, g["Z"], g["noise"], want_dL_dm=True)
    assert info == 0
    check_sparse2(r, g)
    assert np.abs(r["dL_dm"] - g["dL_dm"]).max() <= 1e-6 * np.abs(g["dL_dm"]).max()
    rows = g["rows"]
    B = sctx.fetch_dL_dKnm(int(rows[0]), int(rows[-1] - rows[0] + 1))
    assert np.abs(B[rows - rows[0]] - g["dL_dKnm_rows"]).max() <= 1e-6 * np.abs(g["dL_dKnm_rows"]).max()
    dK = sctx.fetch(L.SparseContext.FETCH_DLDKMM)
    assert np.abs(dK - g["dL_dKmm"]).max() <= 1e-4 * np.abs(g["dL_dKmm"]).max()
    mu, var = sctx.predict(specs, g["Xs"])
    _, cov = sctx.predict(specs, g["Xs"], full_cov=True)
    assert np.abs(mu - g["pred_mu"]).max() <= 1e-6 * np.abs(g["pred_mu"]).max()
    assert np.abs(var - g["pred_var"]).max() <= 1e-5 * np.abs(g["pred_var"]).max()
    assert np.abs(cov - g["pred_cov"]).max() <= 1e-5 * np.abs(g["pred_cov"]).max()


def test_sparse_host_classes_with_add_kernel_hetero_noise_and_mean_function():
    import gpy_amd
    X, Y = O.synthetic(500, 3, seed=21)
    Z = S.synthetic_Z(X, 28, 3)
    # RBF + White, homoscedastic, linear mean function
    w = np.array([[0.2], [-0.1], [0.05]])

    class Mean(object):
        grad = None

        def f(self, Xq):
            return Xq @ w

        def update_gradients(self, dL_dm, Xq):
            Mean.grad = Xq.T @ dL_dm

    k = gpy_amd.RBF(3, variance=1.2, lengthscale=0.9) + gpy_amd.White(3, variance=0.04)
    m = gpy_amd.SparseGP(X, Y, Z, k, gpy_amd.Gaussian(0.12), mean_function=Mean())
    parts = [("rbf", False, 1.2, np.array([0.9]), [0, 1, 2]), ("white", False, 0.04, None, [0, 1, 2])]
    ref = S.vardtc_general(parts, X, Z, Y - X @ w, 0.12)
    assert abs(m.log_likelihood() - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    gref = np.concatenate([ref["dZ"].ravel(), ref["dtheta"], [ref["dnoise"]]])
    assert np.abs(m.gradient - gref).max() <= 1e-6 * np.abs(gref).max()
    assert np.abs(Mean.grad - X.T @ ref["dL_dm"]).max() <= 1e-6 * np.abs(X.T @ ref["dL_dm"]).max()
    Xs = np.random.default_rng(2).standard_normal((40, 3))
    mu, var = m.predict(Xs, include_likelihood=False)
    mur, varr = S.sparse_predict(parts, Z, Xs, ref["woodbury_vector"], ref["woodbury_inv"])
    assert np.abs(mu - (mur + Xs @ w)).max() <= 1e-6 * np.abs(mur).max() and np.abs(var - varr).max() <= 1e-5 * np.abs(varr).max()
    # a foreign consumer materialises dL_dKnm block by block
    G = np.asarray(m.grad_dict["dL_dKnm"])
    assert G.shape == (500, 28) and np.abs(G - ref["dL_dKnm"]).max() <= 1e-6 * np.abs(ref["dL_dKnm"]).max()
    # per-point noise through HeteroscedasticGaussian
    meta = {"output_index": np.arange(500)[:, None]}
    lik = gpy_amd.HeteroscedasticGaussian(meta, variance=0.1)
    lik.variance[:] = 0.05 + 0.1 * np.random.default_rng(3).random(500)
    k2 = gpy_amd.Matern52(3, variance=0.9, lengthscale=[0.8, 1.1, 1.5], ARD=True)
    m2 = gpy_amd.SparseGP(X, Y, Z, k2, lik, Y_metadata=meta)
    ref2 = S.vardtc_general([("matern52", True, 0.9, np.array([0.8, 1.1, 1.5]), [0, 1, 2])], X, Z, Y, lik.variance.values)
    assert abs(m2.log_likelihood() - ref2["lml"]) <= 1e-9 * abs(ref2["lml"])
    assert np.abs(lik.variance.gradient - ref2["dnoise"]).max() <= 1e-6 * np.abs(ref2["dnoise"]).max()
    assert np.abs(k2.gradient - ref2["dtheta"]).max() <= 1e-6 * np.abs(ref2["dtheta"]).max()


def test_sparse_host_classes_with_a_product_kernel():
    """`SparseGP` with `Prod` (reference prod.py:58-113) + White through the drop-in classes: LML, every gradient (Z, both
    factors, White, noise), prediction -- against the oracle (pinned to the reference's own VarDTC on product kernels by the
    sparse3_prod_* fixtures) and by a finite-difference check."""
    import gpy_amd
    X, Y = O.synthetic(420, 3, seed=9)
    Z = S.synthetic_Z(X, 24, 9)
    k = gpy_amd.RBF(2, variance=1.3, lengthscale=0.9, active_dims=[0, 1]) * gpy_amd.Matern32(1, variance=0.8, lengthscale=1.4,
                                                                                           active_dims=[2])
    k = k + gpy_amd.White(3, variance=0.03)
    m = gpy_amd.SparseGP(X, Y, Z, k, gpy_amd.Gaussian(0.1))
    parts = [("rbf", False, 1.3, np.array([0.9]), [0, 1], 1), ("matern32", False, 0.8, np.array([1.4]), [2], 1),
             ("white", False, 0.03, None, [0, 1, 2], 0)]
    ref = S.vardtc_general(parts, X, Z, Y, 0.1)
    assert abs(m.log_likelihood() - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    gref = np.concatenate([ref["dZ"].ravel(), ref["dtheta"], [ref["dnoise"]]])
    assert np.abs(m.gradient - gref).max() <= 1e-6 * np.abs(gref).max()
    Xs = np.random.default_rng(4).standard_normal((30, 3))
    mu, var = m.predict(Xs, include_likelihood=False)
    mur, varr = S.sparse_predict(parts, Z, Xs, ref["woodbury_vector"], ref["woodbury_inv"])
    assert np.abs(mu - mur).max() <= 1e-6 * np.abs(mur).max() and np.abs(var - varr).max() <= 1e-5 * np.abs(varr).max()
    assert m.checkgrad()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_mode_over_the_loopback_transport(world):
    """world > 1 on ONE GPU: `world` contexts, one host thread each, meet at the in-process rendezvous for both exchange
    steps (mi355gp_sparse_attach_loopback).  Every rank must return the unsharded result -- incl. per-point noise, whose
    log-likelihood sums need their own all-reduce."""
    import threading
    from gpy_amd import grid as G
    N, M, D = 5000, 96, 4
    X, Y = O.synthetic(N, D, seed=5)
    Z = S.synthetic_Z(X, M, 0)
    parts = [("rbf", True, 1.1, np.array([0.7, 1.0, 1.3, 1.6]), [0, 1, 2, 3]), ("bias", False, 0.2, None, [0, 1, 2, 3])]
    specs = _specs_of({"parts": parts})
    noise_het = 0.05 + 0.1 * np.random.default_rng(1).random(N)
    for noise in (0.1, noise_het):
        ref = S.vardtc_general(parts, X, Z, Y, noise)
        out, errs = [None] * world, []

        def work(rank):
            try:
                c = L.SparseContext(0)
                try:
                    lo, hi = G.shard_rows(N, rank, world)
                    c.attach_loopback(rank, world, group_key=1000 + world * 10 + (1 if np.ndim(noise) else 0))
                    c.set_data(X[lo:hi], Y[lo:hi])
                    info, r = c.vardtc_sum(specs, Z, noise[lo:hi] if np.ndim(noise) else noise)
                    assert info == 0
                    out[rank] = (r, lo, hi)
                finally:
                    c.close()
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(timeout=120) for t in ts]
        assert not errs, errs
        for r, lo, hi in out:
            assert abs(r["lml"] - ref["lml"]) <= 1e-9 * abs(ref["lml"])
            assert np.abs(r["dtheta"] - ref["dtheta"]).max() <= 1e-6 * np.abs(ref["dtheta"]).max()
            assert np.abs(r["dZ"] - ref["dZ"]).max() <= 1e-6 * np.abs(ref["dZ"]).max()
            if np.ndim(noise):
                assert np.abs(r["dnoise"] - ref["dnoise"][lo:hi]).max() <= 1e-6 * np.abs(ref["dnoise"]).max()
            else:
                assert abs(r["dnoise"] - ref["dnoise"]) <= 1e-6 * abs(ref["dnoise"])
        assert out[0][0]["dtheta"].tobytes() == out[-1][0]["dtheta"].tobytes()      # identical on every rank


_RCCL_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from gpy_amd import _lib as L, grid as G
from oracle import gp_oracle as O, sparse_oracle as S
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
idb = G.unique_id() if rank == 0 else b"\0" * G.ID_BYTES
idb = G.exchange_id_file(idb, rank, %(idfile)r, world=world)
if %(mode)r == "sparse":
    N, M, D = 40000, 256, 4
    X, Y = O.synthetic(N, D, seed=5); Z = S.synthetic_Z(X, M, 0)
    var, ls, noise = O.default_theta(D, True)
    lo, hi = G.shard_rows(N, rank, world)
    c = L.SparseContext(rank); c.attach_comm(rank, world, idb); c.set_data(X[lo:hi], Y[lo:hi])
    info, r = c.vardtc("rbf", True, L.theta_vec(var, ls, True, D), Z, noise)
    res = dict(info=info, lml=r["lml"], dtheta=r["dtheta"].tolist(), dZn=float(np.linalg.norm(r["dZ"])))
else:
    N, D = 6000, 5
    X, Y = O.synthetic(N, D, seed=3)
    var, ls, noise = O.default_theta(D, True)
    g = G.GridContext(rank, rank, world, 1, 2, 256, idb); g.set_data(X, Y)
    info, r = g.exact_inference("matern52", True, L.theta_vec(var, ls, True, D), noise)
    res = dict(info=info, lml=r["lml"], dtheta=r["dtheta"].tolist(), an=float(np.linalg.norm(r["alpha"])))
json.dump(res, open(%(out)r %% rank, "w"))
"""


@pytest.mark.parametrize("mode", ["grid", "sparse"])
def test_two_process_rccl_over_xgmi(mode, tmp_path):
    """e1 / e2 with TWO processes on TWO GPUs over RCCL (panel broadcasts of the 1 x 2 block-cyclic grid; the all-reduces of
    the row-sharded sparse path).  Skips on a single-GPU box; the loopback transports cover the same logic there."""
    import json
    import os
    import subprocess
    import sys
    if L.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_RCCL_WORKER % dict(root=root, idfile=str(tmp_path / "id.bin"), mode=mode, out=str(tmp_path / "r%d.json")))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2",
                                                                     HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2)]
    assert res[0]["info"] == 0 and res[0]["lml"] == res[1]["lml"] and res[0]["dtheta"] == res[1]["dtheta"]
    if mode == "sparse":
        X, Y = O.synthetic(40000, 4, seed=5)
        Z = S.synthetic_Z(X, 256, 0)
        var, ls, noise = O.default_theta(4, True)
        ref = S.vardtc("rbf", X, Z, Y, var, ls, True, noise)
        assert abs(res[0]["lml"] - ref["lml"]) <= 1e-9 * abs(ref["lml"])
        assert np.abs(np.array(res[0]["dtheta"]) - ref["dtheta"]).max() <= 1e-6 * np.abs(ref["dtheta"]).max()
    else:
        X, Y = O.synthetic(6000, 5, seed=3)
        var, ls, noise = O.default_theta(5, True)
        ref = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
        gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
        assert abs(res[0]["lml"] - ref["lml"]) <= 1e-10 * abs(ref["lml"])
        assert np.abs(np.array(res[0]["dtheta"]) - gref).max() <= 1e-8 * np.abs(gref).max()


def test_sparse_integration_stub_against_the_real_library():
    """integration/gpy_mi355x.py `make_sparse_classes` (the reference-side VarDTC binding of INTEGRATION.md) bound to the real
    libmi355gp.so, with gpy_amd's classes standing in for GPy's: posterior, LML, the fused gradients and the row-block
    materialisation of dL_dKnm against the sparse oracle (the CPU suite runs the same file against the reference's own
    classes with the library mocked by the oracle: tests/test_integration_stub.py)."""
    import os
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration"))
    import gpy_amd
    import gpy_mi355x
    from oracle import sparse_oracle as SO
    from gpy_amd.sparse import SparsePosterior
    lib = gpy_mi355x.bind_sparse(gpy_mi355x.bind(L.LIB_PATH))
    ns = types.SimpleNamespace(RBF=gpy_amd.RBF, Matern52=gpy_amd.Matern52, Matern32=gpy_amd.Matern32,
                               Exponential=gpy_amd.Exponential, PosteriorExact=gpy_amd.PosteriorExact,
                               LatentFunctionInference=object,
                               Posterior=lambda woodbury_inv, woodbury_vector, K, mean, cov, K_chol: SparsePosterior(
                                   woodbury_inv, woodbury_vector, K, K_chol))
    C = gpy_mi355x.make_classes(lib, ns)
    SC = gpy_mi355x.make_sparse_classes(lib, ns)
    X, Y = O.synthetic(9000, 4, seed=21)                   # more than one 8192-row block of dL_dKnm
    Z = SO.synthetic_Z(X, 70, 3)
    var, ls, noise = O.default_theta(4, True)
    ref = SO.vardtc("rbf", X, Z, Y, var, ls, True, noise)
    k = C.RBF(4, variance=var, lengthscale=ls, ARD=True)
    post, lml, gd = SC.VarDTC().inference(k, X, Z, gpy_amd.Gaussian(variance=noise), Y)
    assert abs(lml - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    assert np.abs(post.woodbury_vector - ref["woodbury_vector"]).max() <= 1e-6 * np.abs(ref["woodbury_vector"]).max()
    assert np.abs(gd["fused"]["dtheta"] - ref["dtheta"]).max() <= 1e-6 * np.abs(ref["dtheta"]).max()
    assert np.abs(gd["fused"]["dZ"] - ref["dZ"]).max() <= 1e-6 * np.abs(ref["dZ"]).max()
    assert abs(gd["dL_dthetaL"] - ref["dnoise"]) <= 1e-6 * abs(ref["dnoise"])
    G = np.asarray(gd["dL_dKnm"])
    assert G.shape == (9000, 70) and np.abs(G - ref["dL_dKnm"]).max() <= 1e-6 * np.abs(ref["dL_dKnm"]).max()
    assert np.abs(np.asarray(gd["dL_dKmm"]) - ref["dL_dKmm"]).max() <= 1e-4 * np.abs(ref["dL_dKmm"]).max()

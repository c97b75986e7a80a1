"""GPU (-m gpu): the 2D block-cyclic multi-GPU mode (csrc/grid.hip) with all Pr*Pc logical ranks on ONE device
(LOOPBACK transport), against the CPU oracle and the single-GPU path; plus the RCCL transport at world size 1.
Tolerances: the fp64 parity contract of the single-GPU tests."""
import numpy as np
import pytest

from conftest import diag_lib
from gpy_amd import _lib as L
from gpy_amd import grid as G
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
TOL_LML, TOL_ALPHA, TOL_GRAD = 1e-10, 1e-9, 1e-8


def _check(res, ref, noise_scalar=True):
    assert abs(res["lml"] - ref["lml"]) <= TOL_LML * max(1.0, abs(ref["lml"]))
    assert np.linalg.norm(res["alpha"] - ref["alpha"]) <= TOL_ALPHA * np.linalg.norm(ref["alpha"])
    gref = np.concatenate([[ref["dvar"]], np.atleast_1d(ref["dlen"])])
    assert np.abs(res["dtheta"] - gref).max() <= TOL_GRAD * np.abs(gref).max()
    if noise_scalar:
        assert abs(res["dnoise"] - ref["dL_dnoise"]) <= TOL_GRAD * abs(ref["dL_dnoise"])


CASES = [
    # kind, ARD, N, D, Dy, Pr, Pc, nb
    ("rbf", False, 300, 3, 1, 1, 1, 128),
    ("matern52", True, 500, 4, 2, 1, 1, 256),
    ("rbf", False, 700, 4, 1, 2, 2, 128),
    ("matern52", True, 1000, 5, 2, 2, 4, 128),
    ("matern32", True, 900, 3, 1, 3, 2, 256),
    ("exponential", False, 777, 2, 1, 1, 3, 256),
    ("rbf", True, 1500, 8, 1, 4, 2, 128),
    ("matern52", False, 2048, 8, 1, 2, 2, 512),
]


def _generic_1x1(on):
    """A 1 x 1 loopback grid degenerates to the single-GPU pipeline; MI355GP_GRID_FORCE_GENERIC=1 (diagnostics build only) keeps
    the generic one-pass code of the grid mode, so that it is covered at world size 1 too."""
    import os
    if on:
        os.environ["MI355GP_GRID_FORCE_GENERIC"] = "1"
    else:
        os.environ.pop("MI355GP_GRID_FORCE_GENERIC", None)


@pytest.mark.parametrize("kind,ARD,N,D,Dy,Pr,Pc,nb", CASES)
def test_loopback_grid_matches_oracle(kind, ARD, N, D, Dy, Pr, Pc, nb):
    _loopback_case(kind, ARD, N, D, Dy, Pr, Pc, nb)


@diag_lib
def test_loopback_1x1_grid_on_the_generic_code_matches_oracle():
    _generic_1x1(True)
    try:
        _loopback_case("rbf", False, 300, 3, 1, 1, 1, 128)
    finally:
        _generic_1x1(False)


def _loopback_case(kind, ARD, N, D, Dy, Pr, Pc, nb):
    X, Y = O.synthetic(N, D, seed=N + Pr, Dy=Dy)
    var, ls, noise = O.default_theta(D, ARD)
    ref = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
    g = G.GridContext.loopback(Pr, Pc, nb)
    try:
        g.set_data(X, Y)
        th = L.theta_vec(var, ls, ARD, D)
        info, res = g.exact_inference(kind, ARD, th, noise, want_diag=True, want_stage_ms=True)
        assert info == 0
        _check(res, ref)
        assert abs(res["logdet"] - ref["logdet"]) <= 1e-11 * max(1.0, abs(ref["logdet"]))
        dref = np.diag(ref["dL_dK"])
        assert np.abs(res["diag_dL_dK"] - dref).max() <= TOL_GRAD * np.abs(dref).max()
        Lg = g.fetch(G.FETCH_L)
        assert np.linalg.norm(Lg - ref["L"]) <= 1e-11 * np.linalg.norm(ref["L"])
        Xg = g.fetch(G.FETCH_LINV)
        eye = Xg @ ref["L"]
        assert np.abs(eye - np.eye(N)).max() <= 1e-9
        # second call on the same context (buffers are re-initialised every evaluation)
        info2, res2 = g.exact_inference(kind, ARD, th, noise)
        assert info2 == 0 and res2["lml"] == res["lml"] and np.array_equal(res2["dtheta"], res["dtheta"])
    finally:
        g.close()


@pytest.mark.parametrize("Pr,Pc,nb,N", [(2, 4, 128, 1150), (3, 2, 128, 900), (2, 2, 256, 2000)])
def test_two_level_blocking_options(Pr, Pc, nb, N):
    """The group size G of the two-level blocked factorisation, the W = X^T X aggregation GW and the look-ahead switch
    (mi355gp_grid_set_option) change the schedule, never the result beyond rounding: every combination against the oracle,
    including groups that do not divide the number of steps and G larger than the number of steps."""
    _blocking_options_case(Pr, Pc, nb, N)


@diag_lib
def test_two_level_blocking_options_on_a_1x1_grid_with_the_generic_code():
    _generic_1x1(True)
    try:
        _blocking_options_case(1, 1, 128, 700)
    finally:
        _generic_1x1(False)


def _blocking_options_case(Pr, Pc, nb, N):
    kind, ARD, D = "matern52", True, 4
    X, Y = O.synthetic(N, D, seed=N + 3, Dy=2)
    var, ls, noise = O.default_theta(D, ARD)
    ref = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
    g = G.GridContext.loopback(Pr, Pc, nb)
    try:
        assert g.get_option("G") == 1 and g.get_option("GW") == 4 and g.get_option("lookahead") == 1
        g.set_data(X, Y)
        th = L.theta_vec(var, ls, ARD, D)
        for Gs, GW, la in [(1, 1, 1), (1, 0, 0), (2, 0, 1), (2, 3, 1), (3, 2, 0), (4, 0, 1), (4, 4, 1), (5, 1, 1), (64, 0, 1),
                           (64, 7, 0)]:
            g.set_option("G", Gs)
            g.set_option("GW", GW)
            g.set_option("lookahead", la)
            assert (g.get_option("G"), g.get_option("GW"), g.get_option("lookahead")) == (Gs, GW, la)
            info, res = g.exact_inference(kind, ARD, th, noise, want_diag=True)
            assert info == 0, (Gs, GW, la)
            _check(res, ref)
            dref = np.diag(ref["dL_dK"])
            assert np.abs(res["diag_dL_dK"] - dref).max() <= TOL_GRAD * np.abs(dref).max()
            Lg = g.fetch(G.FETCH_L)
            assert np.linalg.norm(Lg - ref["L"]) <= 1e-11 * np.linalg.norm(ref["L"])
            Xg = g.fetch(G.FETCH_LINV)
            assert np.abs(Xg @ ref["L"] - np.eye(N)).max() <= 1e-9
        g.set_option("G", -1)
        assert g.get_option("G") == 1
    finally:
        g.close()


def test_grid_agrees_with_single_gpu_path_and_heteroscedastic_noise():
    N, D = 1100, 6
    X, Y = O.synthetic(N, D, seed=5)
    var, ls, _ = O.default_theta(D, True)
    noise = 0.05 + 0.1 * np.random.default_rng(1).random(N)
    th = L.theta_vec(var, ls, True, D)
    c = L.Context(0)
    g = G.GridContext.loopback(2, 3, 256)
    try:
        c.set_data(X, Y)
        i1, r1 = c.exact_inference("matern52", True, th, noise, want_diag=True)
        g.set_data(X, Y)
        i2, r2 = g.exact_inference("matern52", True, th, noise, want_diag=True)
        assert i1 == 0 and i2 == 0
        assert abs(r1["lml"] - r2["lml"]) <= TOL_LML * abs(r1["lml"])
        assert np.linalg.norm(r1["alpha"] - r2["alpha"]) <= TOL_ALPHA * np.linalg.norm(r1["alpha"])
        assert np.abs(r1["dtheta"] - r2["dtheta"]).max() <= TOL_GRAD * np.abs(r1["dtheta"]).max()
        assert np.abs(r1["diag_dL_dK"] - r2["diag_dL_dK"]).max() <= TOL_GRAD * np.abs(r1["diag_dL_dK"]).max()
    finally:
        c.close()
        g.close()


def test_grid_reports_not_positive_definite_on_every_rank():
    N, D = 400, 2
    X, Y = O.synthetic(N, D, seed=3)
    X[200:] = X[:200]                      # duplicated points, no noise -> singular
    g = G.GridContext.loopback(2, 2, 128)
    try:
        g.set_data(X, Y)
        info, _ = g.exact_inference("rbf", False, L.theta_vec(1.0, 1.0, False, D), 0.0, jitter=0.0)
        assert info > 0
        info, res = g.exact_inference("rbf", False, L.theta_vec(1.0, 1.0, False, D), 0.0, jitter=0.0, extra_jitter=1e-4)
        assert info == 0 and np.isfinite(res["lml"])
    finally:
        g.close()


def test_rccl_transport_world_size_one():
    """The RCCL code path (dlopen, ncclCommInitRank, ncclCommSplit, broadcasts, all-reduces) on the one GPU we have."""
    N, D = 600, 3
    X, Y = O.synthetic(N, D, seed=9)
    var, ls, noise = O.default_theta(D, False)
    ref = O.parameters_changed("rbf", X, Y, var, ls, False, noise)
    g = G.GridContext(0, 0, 1, 1, 1, 256, G.unique_id())
    try:
        assert not g.is_loopback
        g.set_data(X, Y)
        info, res = g.exact_inference("rbf", False, L.theta_vec(var, ls, False, D), noise)
        assert info == 0
        _check(res, ref)
    finally:
        g.close()

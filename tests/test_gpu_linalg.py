"""GPU (-m gpu): `gpy_amd.linalg.jitchol` / `pdinv` -- the device-backed mirror of `GPy/util/linalg.py:56-75,193-214` -- against
the oracle's restatement (itself pinned to the reference's own functions by tests/test_oracle_vs_reference.py): values, the
4-tuple layout, the jitter ladder and the two LinAlgError messages."""
import numpy as np
import pytest

from conftest import DIAG_LIB

import gpy_amd
from gpy_amd import _lib as L
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _spd(n, seed, cond_shift=0.5):
    X, _ = O.synthetic(n, 3, seed=seed)
    return O.kern_K("matern52", X, None, 1.3, np.array([0.8, 1.1, 1.9]), True) + cond_shift * np.eye(n)


@pytest.mark.parametrize("n", [1, 17, 128, 700, 1411])
def test_pdinv_tuple_matches_oracle(n):
    A = _spd(n, seed=n)
    Ai, L, Li, logdet = gpy_amd.linalg.pdinv(A)
    rAi, rL, rLi, rlogdet = O.pdinv(A)
    assert abs(logdet - rlogdet) <= 1e-12 * max(1.0, abs(rlogdet))
    assert np.abs(L - rL).max() <= 1e-12 * np.abs(rL).max() and np.all(np.triu(L, 1) == 0.0)
    assert np.abs(Li - rLi).max() <= 1e-11 * np.abs(rLi).max() and np.all(np.triu(Li, 1) == 0.0)
    assert np.abs(Ai - rAi).max() <= 1e-11 * np.abs(rAi).max() and np.array_equal(Ai, Ai.T)
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max() * n
    assert Ai.flags["C_CONTIGUOUS"] and gpy_amd.util.linalg.pdinv is gpy_amd.linalg.pdinv


def test_jitchol_ladder_and_errors_follow_the_reference():
    rng = np.random.default_rng(0)
    B = rng.standard_normal((300, 40))
    A = B @ B.T                                          # rank 40: dpotrf fails, the ladder adds mean(diag) * 1e-6 ...
    L = gpy_amd.linalg.jitchol(A)
    Lr = O.jitchol(A)                                    # the oracle runs the same ladder (util/linalg.py:61-75)
    # the same jitter level was accepted: L L^T differs from A by exactly that multiple of the identity
    d = np.diag(L @ L.T - A)
    dr = np.diag(Lr @ Lr.T - A)
    assert np.allclose(d, d.mean(), rtol=1e-6) and np.isclose(d.mean(), dr.mean(), rtol=1e-6)
    assert np.all(np.triu(L, 1) == 0.0)
    with pytest.raises(np.linalg.LinAlgError, match="not pd: non-positive diagonal elements"):
        gpy_amd.linalg.jitchol(np.array([[1.0, 2.0], [2.0, -1.0]]))
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite, even with jitter."):
        gpy_amd.linalg.jitchol(np.array([[1.0, 1e3], [1e3, 1.0]]))
    # a positive-definite matrix passes through untouched
    A2 = _spd(200, seed=3)
    assert np.abs(gpy_amd.linalg.jitchol(A2) - O.jitchol(A2)).max() <= 1e-12


@pytest.mark.parametrize("N", [256, 384, 640, 1000, 1408, 2048, 2944, 3333, 4096, 4224, 4608, 5120, 5248, 6500, 8192])
def test_persistent_dataflow_cholesky_is_bit_identical_to_the_launch_per_step_schedule(N):
    """persist.hip: one persistent launch (chain workgroup + static tile owners, write-through hand-offs between workgroups on
    different XCDs) against factor.hip's launch-per-step schedule on the same resident SPD matrix: every double of the lower
    triangle of L must have the same BITS, no wait may have timed out, and the chain's timeline must be monotone.  The sizes cross
    the policy boundaries of round 6: the second sub-diagonal in 64-row halves up to nt = 40 (5120 / 5248), far tiles dealt out
    column-major from nt = 33 (4096 / 4224), odd tile counts; `tools/persist_all_sizes.py` runs every nt = 2 .. 64."""
    r = L.dbg_persist(N, reps=2)
    assert r["mismatches"] == 0 and r["info"] == 0 and r["abort"] == 0
    st = r["steps"]
    assert np.all(np.diff(st[:, 0]) > 0) and np.all(st[:, 1] > st[:, 0])       # factor(j) starts after factor(j-1), ends after it starts
    assert r["ms_persist"] < 2.0 * r["ms_steps"] + 0.05


@pytest.mark.parametrize("N", [6400, 7300, 8192])
def test_paired_far_updates_are_bit_identical_to_one_panel_per_pass(N):
    """Option "agg2": from N = 6144 the look-ahead schedule applies two adjacent panels to the far columns in ONE pass over C
    (K = 1024).  Every tile still receives its panels in ascending order with the accumulator carried in fp64 through C, so the
    result has the same BITS as the one-panel-per-pass schedule and as the in-order schedule (look-ahead off); an even and
    an odd number of panels and a ragged last panel are covered."""
    X, Y = O.synthetic(N, 4, seed=N)
    var, ls, noise = O.default_theta(4, True)
    th = L.theta_vec(var, ls, True, 4)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        assert c.get_option("agg2") == 0
        c.set_option("persist", 0)                           # (from round 6 the persistent launch would take these sizes)
        outs = []
        for agg, la in ((1, 1), (0, 1), (1, 0), (1, 1)):
            c.set_option("agg2", agg)
            c.set_option("lookahead", la)
            info, r = c.exact_inference("matern52", True, th, noise)
            assert info == 0
            outs.append((r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes()))
        assert outs[0] == outs[1] == outs[2] == outs[3]
    finally:
        c.close()


def test_persistent_cholesky_option_reports_non_pd_and_can_be_switched_off_per_context():
    """The product path takes the persistent launch up to N = 8192 (`FACTOR_PERSIST_MAX_NT`): option "persist" = 1 (default)
    runs the factorisation as one launch, 0 returns a context to the launch-per-step schedule.  Both give the same bits and
    report the same LAPACK-style info on a non-PD matrix."""
    X, Y = O.synthetic(1500, 3, seed=5)
    var, ls, noise = O.default_theta(3, False)
    th = L.theta_vec(var, ls, False, 3)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        assert c.get_option("persist") == 1
        outs = []
        for p in (1, 0, 1):
            c.set_option("persist", p)
            info, r = c.exact_inference("rbf", False, th, noise)
            assert info == 0
            outs.append((r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes()))
        assert outs[0] == outs[1] == outs[2] and c.get_option("persist_aborts") == 0
        # duplicated inputs, no noise, negative jitter: the Gram matrix is not positive definite -> same info either way
        Xd = np.vstack([X[:700], X[:700], X[:100]])
        c.set_data(Xd, Y)
        infos = []
        for p in (2, 1, 0):
            c.set_option("persist", p)
            info, _ = c.exact_inference("rbf", False, th, 0.0, jitter=-1e-3)
            infos.append(info)
        assert infos[0] == infos[1] == infos[2] and infos[0] > 0
    finally:
        c.close()


def test_a_context_settles_on_one_schedule_by_its_own_timing_and_reports_it():
    """`MI355GP_PERSIST_AUTO`: from nt = 16 on a context times its fourth and fifth evaluation (persistent launch
    + early inverse, warm), then runs three on launches (the first untimed), compares the minima, keeps the faster schedule and says which through the read-only option "persist_sched"
    (0 undecided, 1 persistent launch, 2 launches).  Every evaluation on the way has the same bits; an explicit "persist"
    option ends the calibration, -1 re-opens it."""
    X, Y = O.synthetic(2304, 4, seed=11)
    var, ls, noise = O.default_theta(4, False)
    th = L.theta_vec(var, ls, False, 4)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        assert c.get_option("persist_sched") == 0
        outs = []
        for _ in range(9):                                     # 3 + two persistent samples + one untimed and two timed on launches (+1)
            info, r = c.exact_inference("rbf", False, th, noise)
            assert info == 0
            outs.append((r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes()))
        assert all(o == outs[0] for o in outs)
        assert c.get_option("persist_sched") in (1, 2) and c.get_option("persist_aborts") == 0
        with pytest.raises(L.MI355GPError):
            c.set_option("persist_sched", 1)
        c.set_option("persist", 0)
        assert c.get_option("persist_sched") == 2
        c.set_option("persist", 1)
        assert c.get_option("persist_sched") == 1
        c.set_option("persist", -1)
        assert c.get_option("persist_sched") == 0
        info, r = c.exact_inference("rbf", False, th, noise)
        assert info == 0 and (r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes()) == outs[0]
    finally:
        c.close()


def test_one_shot_factorisations_follow_the_box_verdict_with_the_same_bits(tmp_path):
    """About half of the boxes run the persistent launch at half speed; contexts find out by timing both schedules, and the
    one-shot entry points (pdinv / jitchol: a fresh workspace per call) follow the process's verdict for machine-filling sizes.
    Either schedule gives the same L and log-determinant bits: two fresh processes, one starting from each verdict."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import gpy_amd; from oracle import gp_oracle as O\n"
            "X, _ = O.synthetic(3000, 3, seed=4)\n"
            "A = O.kern_K('matern52', X, None, 1.3, np.array([0.8, 1.1, 1.9]), True) + 0.5 * np.eye(3000)\n"
            "Ai, L, Li, ld = gpy_amd.linalg.pdinv(A)\n"
            "np.save(sys.argv[1], L); print(repr(float(ld)))\n") % root
    outs = []
    for verdict in ("0", "1"):
        f = str(tmp_path / ("L%s.npy" % verdict))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, MI355GP_DBG_BOX_VERDICT=verdict, MI355GP_LIB=DIAG_LIB),   # (a switch of the diagnostics build)
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(f), r.stdout.strip().splitlines()[-1]))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]

"""GPU (-m gpu): the persistent single-launch Cholesky (persist.hip) is the default up to N = 8192 and spin-waits between
workgroups, so it needs all its workgroups resident at once.  What happens when they are not: the launch is called off at
its co-residency gate with the matrix untouched (or, never seen in a sane run, aborts on a wait timeout) and the SAME C-ABI
call redoes the factorisation on the launch-per-step schedule -- same bits, no error, nothing for the jitter ladder to see
(`GPy/util/linalg.py:56-75` must only ever hear of genuine non-positive pivots).  Covered: fault injection through
`MI355GP_OPT_PERSIST_TEST` (exact path), the dense `pdinv` / `jitchol` entry points, both M x M factorisations of VarDTC,
two host threads with a context each, and two PROCESSES sharing the one GPU."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import diag_lib

import gpy_amd
from gpy_amd import _lib as L
from oracle import gp_oracle as O
from oracle import sparse_oracle as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _key(r):
    return (r["lml"], r["dtheta"].tobytes(), r["alpha"].tobytes())


def _close(r, r0):
    """equal to rounding (the persistent launch and the launch-per-step schedule it falls back to give the same bits; the
    looser check keeps the safety tests independent of that)"""
    return (abs(r["lml"] - r0["lml"]) <= 1e-12 * abs(r0["lml"]) and
            np.abs(r["alpha"] - r0["alpha"]).max() <= 1e-11 * np.abs(r0["alpha"]).max() and
            np.abs(r["dtheta"] - r0["dtheta"]).max() <= 1e-10 * np.abs(r0["dtheta"]).max())


@pytest.mark.parametrize("mode", [1, 2])
@diag_lib
def test_called_off_or_aborted_persistent_launch_is_redone_inside_the_call(mode):
    """mode 1: the launch waits for a workgroup that never comes -> called off after 1 ms, matrix untouched; mode 2: the chain
    workgroup gives up after the gate -> the workers time out on it, the matrix is rebuilt.  Either way the call returns the
    bits of an undisturbed evaluation; a clean call-off is retried after PS_SKIP_AFTER_CLEAN evaluations, a dirty abort
    keeps the context on the launch-per-step schedule."""
    N = 2048
    X, Y = O.synthetic(N, 5, seed=21)
    var, ls, noise = O.default_theta(5, True)
    th = L.theta_vec(var, ls, True, 5)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        info, r0 = c.exact_inference("matern52", True, th, noise)
        assert info == 0 and c.get_option("persist_aborts") == 0
        for _ in range(3):                                    # plain, capture, replay: the graph is live
            info, r = c.exact_inference("matern52", True, th, noise)
            assert info == 0 and _key(r) == _key(r0)
        c.set_option("persist_test", mode)
        info, rs = c.exact_inference("matern52", True, th, noise)       # redone on launches inside the call
        assert info == 0 and _close(rs, r0)
        assert c.get_option("persist_aborts") == 1
        skip = c.get_option("persist_skip")
        assert skip > 0
        if mode == 1:
            for it in range(skip + 4):                        # launches while the skip lasts, then the persistent schedule again
                info, r = c.exact_inference("matern52", True, th, noise)
                assert info == 0 and _key(r) == (_key(rs) if it < skip else _key(r0)), it
            assert c.get_option("persist_skip") == 0 and c.get_option("persist_aborts") == 1
        else:
            for _ in range(3):
                info, r = c.exact_inference("matern52", True, th, noise)
                assert info == 0 and _key(r) == _key(rs)
            assert c.get_option("persist_skip") == skip       # stays off
        # a genuinely non-PD matrix still reports its LAPACK info through the redone factorisation
        Xd = np.vstack([X[:900], X[:900], X[:248]])
        c.set_data(Xd, Y)
        info_ref, _ = c.exact_inference("matern52", True, th, 0.0, jitter=-1e-3)
        c.set_option("persist_test", 1)
        info_inj, _ = c.exact_inference("matern52", True, th, 0.0, jitter=-1e-3)
        assert info_ref > 0 and info_inj == info_ref
    finally:
        c.close()


@diag_lib
def test_a_gate_of_the_early_inverse_that_gives_up_voids_the_evaluation():
    """The inverse of the leading half runs on the side stream underneath the persistent launch, behind one-thread gate kernels
    that wait on the launch's progress words.  A gate that gives up while the launch is still running (a GPU shared with
    something heavy: 20 ms limit) lets its consumers read rows of L that are not final: it must mark the evaluation as aborted
    so that the host redoes it on launches.  Fault injection (persist_test = 3: the gate gives up at once): the call returns
    the undisturbed result, counts one abort, and the context stays on launches."""
    N = 4096
    X, Y = O.synthetic(N, 4, seed=33)
    var, ls, noise = O.default_theta(4, False)
    th = L.theta_vec(var, ls, False, 4)
    c = L.Context(0)
    try:
        c.set_data(X, Y)
        c.set_option("persist", 1)                            # an explicit choice: no calibration by measurement in this test
        outs = []
        for _ in range(3):                                    # the early inverse starts with the second evaluation
            info, r = c.exact_inference("rbf", False, th, noise)
            assert info == 0
            outs.append(_key(r))
        assert outs[0] == outs[1] == outs[2] and c.get_option("persist_aborts") == 0
        c.set_option("persist_test", 3)
        info, rs = c.exact_inference("rbf", False, th, noise)
        assert info == 0 and _key(rs) == outs[0]              # redone on launches: the same bits
        assert c.get_option("persist_aborts") == 1 and c.get_option("persist_skip") > 0
        info, r = c.exact_inference("rbf", False, th, noise)
        assert info == 0 and _key(r) == outs[0]
    finally:
        c.close()


@pytest.mark.parametrize("mode", [1, 2])
@diag_lib
def test_dense_pdinv_redoes_a_called_off_launch(mode, monkeypatch):
    n = 1411
    X, _ = O.synthetic(n, 3, seed=9)
    A = O.kern_K("matern52", X, None, 1.3, np.array([0.8, 1.1, 1.9]), True) + 0.5 * np.eye(n)
    ref = gpy_amd.linalg.pdinv(A)
    monkeypatch.setenv("MI355GP_DENSE_PERSIST_TEST", str(mode))
    got = gpy_amd.linalg.pdinv(A)
    assert np.array_equal(ref[1], got[1]) and ref[3] == got[3]            # L and logdet: the same bits on every schedule
    for a, b in ((ref[0], got[0]), (ref[2], got[2])):                      # Ky^-1, L^-1: another summation order
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    Lj = gpy_amd.linalg.jitchol(A)
    assert np.array_equal(Lj, ref[1])


@pytest.mark.parametrize("inject", [1, 2, 11, 12])
@diag_lib
def test_vardtc_redoes_either_m_by_m_factorisation_in_place(inject, monkeypatch):
    """Kmm's launch (1 / 2) or B's (11 / 12) called off / aborted: the factorisation is redone right there (no second pass
    over the data, no collective), the evaluation has the bits of an undisturbed one."""
    N, M, D = 6000, 640, 4
    X, Y = O.synthetic(N, D, seed=4)
    Z = S.synthetic_Z(X, M, 0)
    var, _, noise = O.default_theta(D, False)
    ls = np.array([0.45])                      # a well-conditioned Kmm: the two schedules round L^-1 differently, and the
    th = L.theta_vec(var, ls, False, D)        # difference is amplified by cond(Kmm) (1e-7 relative at the default lengthscale)
    c = L.SparseContext(0)
    try:
        c.set_data(X, Y)
        info, r0 = c.vardtc("rbf", False, th, Z, noise)
        assert info == 0
        monkeypatch.setenv("MI355GP_SPARSE_PERSIST_TEST", str(inject))
        info, r1 = c.vardtc("rbf", False, th, Z, noise)
        monkeypatch.delenv("MI355GP_SPARSE_PERSIST_TEST")
        assert info == 0
        assert abs(r1["lml"] - r0["lml"]) <= 1e-10 * abs(r0["lml"])
        assert np.abs(r1["dtheta"] - r0["dtheta"]).max() <= 1e-7 * np.abs(r0["dtheta"]).max()
        assert np.abs(r1["dZ"] - r0["dZ"]).max() <= 1e-7 * np.abs(r0["dZ"]).max()
        ref = S.vardtc("rbf", X, Z, Y, var, ls, False, noise)
        assert abs(r1["lml"] - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    finally:
        c.close()


def test_two_host_threads_with_a_context_each_at_n4096():
    """Two Python threads, one context each, 50 evaluations of configs[1]'s size each, concurrently (ctypes releases the GIL
    inside the C-ABI call): no error, every evaluation has the bits of a single-threaded one (of the persistent schedule, or
    of the launch-per-step schedule if a launch was called off)."""
    N, D = 4096, 8
    var, ls, noise = O.default_theta(D, False)
    th = L.theta_vec(var, ls, False, D)
    data = [O.synthetic(N, D, seed=s) for s in (0, 1)]
    refs = []
    for X, Y in data:
        c = L.Context(0)
        c.set_data(X, Y)
        both = set()
        for p in (1, 0):                                      # the bits of the persistent schedule and of its fall-back
            c.set_option("persist", p)
            info, r = c.exact_inference("rbf", False, th, noise)
            assert info == 0
            both.add(_key(r))
        refs.append(both)
        c.close()
    errors, aborts = [], [0, 0]

    def work(i):
        try:
            c = L.Context(0)
            c.set_data(*data[i])
            for _ in range(50):
                info, r = c.exact_inference("rbf", False, th, noise)
                if info != 0 or _key(r) not in refs[i]:
                    errors.append((i, info))
            aborts[i] = c.get_option("persist_aborts")
            c.close()
        except Exception as e:                                # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


_CHILD = r"""
import sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from gpy_amd import _lib as L
from gpy_amd.datasets import default_theta, synthetic
N, D, seed = 4096, 8, int(sys.argv[1])
X, Y = synthetic(N, D, seed=seed)
var, ls, noise = default_theta(D, False)
th = L.theta_vec(var, ls, False, D)
c = L.Context(0)
c.set_data(X, Y)
keys = set()
for _ in range(50):
    info, r = c.exact_inference("rbf", False, th, noise)
    assert info == 0
    keys.add((float(r["lml"]).hex(), r["dtheta"].tobytes().hex(), r["alpha"].tobytes().hex()[:256]))
print(json.dumps({"keys": len(keys), "lml": float(r["lml"]).hex(), "dtheta": r["dtheta"].tobytes().hex(),
                  "aborts": c.get_option("persist_aborts")}))
c.close()
"""


def test_two_processes_sharing_one_gpu_at_n4096():
    """Two PROCESSES on the one GPU, each 50 evaluations at N = 4096 on the default (persistent) schedule.  Their persistent
    launches cannot be co-resident (155 KB of LDS per workgroup, one per CU): whichever comes second waits at the gate, and if
    the two interleave both are called off and redo on launches.  No error, no -6, and every evaluation of both processes has
    the bits a process that has the GPU to itself produces (on the persistent schedule or on its fall-back)."""
    import json
    from gpy_amd.datasets import default_theta, synthetic
    N, D = 4096, 8
    var, ls, noise = default_theta(D, False)
    th = L.theta_vec(var, ls, False, D)
    solo = {}
    for seed in (0, 1):
        X, Y = synthetic(N, D, seed=seed)
        c = L.Context(0)
        c.set_data(X, Y)
        solo[seed] = set()
        for p in (1, 0):
            c.set_option("persist", p)
            info, r = c.exact_inference("rbf", False, th, noise)
            assert info == 0
            solo[seed].add((float(r["lml"]).hex(), r["dtheta"].tobytes().hex()))
        c.close()
    code = _CHILD % {"root": ROOT}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for seed in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    for seed, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, se[-2000:]
        rec = json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1])
        assert rec["keys"] <= 2, "more than the two schedules' bit patterns"
        assert (rec["lml"], rec["dtheta"]) in solo[seed]

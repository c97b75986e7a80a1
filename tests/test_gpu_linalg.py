"""GPU (-m gpu): `gpy_amd.linalg.jitchol` / `pdinv` -- the device-backed mirror of `GPy/util/linalg.py:56-75,193-214` -- against
the oracle's restatement (itself pinned to the reference's own functions by tests/test_oracle_vs_reference.py): values, the
4-tuple layout, the jitter ladder and the two LinAlgError messages."""
import numpy as np
import pytest

import gpy_amd
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

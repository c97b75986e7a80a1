"""CPU, only where /root/reference exists: the oracle against the reference's UNMODIFIED files executed through
oracle/ref_loader.py, on fresh seeded inputs (not the golden cases)."""
import numpy as np
import pytest

from oracle import gp_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present (GPU box)")


@pytest.mark.parametrize("kind", O.KINDS)
@pytest.mark.parametrize("ARD", [False, True])
def test_oracle_vs_live_reference(kind, ARD, oracle_native_built):
    ns = ref_loader.load()
    N, D = 257, 4
    X, Y = O.synthetic(N, D, seed=42, Dy=2)
    var, ls, noise = O.default_theta(D, ARD)
    ref = ref_loader.run_iteration(ns, kind, X, Y, var, ls if ARD else float(ls[0]), ARD, noise)
    r = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
    assert abs(r["lml"] - ref["lml"]) <= 1e-12 * abs(ref["lml"])
    np.testing.assert_allclose(r["alpha"], ref["alpha"], rtol=0, atol=1e-12 * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(r["L"], ref["L"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r["dL_dK"], ref["dL_dK"], rtol=0, atol=1e-12 * np.abs(ref["dL_dK"]).max())
    np.testing.assert_allclose(r["dvar"], ref["dvar"][0], rtol=1e-11)
    np.testing.assert_allclose(r["dlen"], ref["dlen"], rtol=1e-10)
    np.testing.assert_allclose(r["dL_dnoise"], ref["dnoise"][0], rtol=1e-11)


def test_reference_kernels_vs_oracle_cross():
    ns = ref_loader.load()
    rng = np.random.default_rng(9)
    X, X2 = rng.standard_normal((50, 3)), rng.standard_normal((20, 3))
    for kind in O.KINDS:
        k = ref_loader.make_kernel(ns, kind, 3, 1.7, [0.5, 1.0, 2.0], True)
        np.testing.assert_allclose(O.kern_K(kind, X, X2, 1.7, [0.5, 1.0, 2.0], True), k.K(X, X2), rtol=0, atol=1e-15)
        G = rng.standard_normal((50, 20))
        k.update_gradients_full(G, X, X2)
        dv, dl = O.update_gradients_full(kind, G, X, X2, 1.7, [0.5, 1.0, 2.0], True)
        np.testing.assert_allclose(dv, k.variance.gradient[0], rtol=1e-12)
        np.testing.assert_allclose(dl, k.lengthscale.gradient, rtol=1e-11)

"""CPU: pins the full-size BASELINE fixtures (tests/golden/baseline_*.npz, generated from the reference by
oracle/make_golden_baseline.py) and the lean large-N oracle that produced the N=32768 one."""
import numpy as np
import pytest

from conftest import baseline_golden
from oracle import gp_oracle as O


def test_oracle_matches_reference_golden_at_config2_full_size():
    """BASELINE configs[1] (RBF, N=4096, D=8) in full: the NumPy/SciPy oracle against the reference's own outputs."""
    g = baseline_golden("baseline_c2_rbf_n4096_d8")
    X, Y = O.synthetic(g["N"], g["D"], seed=g["seed"])
    r = O.parameters_changed(g["kind"], X, Y, g["variance"], g["lengthscale"], g["ARD"], g["noise"])
    assert abs(r["lml"] - g["lml"]) <= 1e-12 * abs(g["lml"])
    assert np.linalg.norm(r["alpha"] - g["alpha"]) <= 1e-10 * np.linalg.norm(g["alpha"])
    assert abs(r["dvar"] - g["dvar"][0]) <= 1e-9 * abs(g["dvar"][0])
    assert np.abs(r["dlen"] - g["dlen"]).max() <= 1e-9 * np.abs(g["dlen"]).max()
    assert abs(r["dL_dnoise"] - g["dnoise"][0]) <= 1e-9 * abs(g["dnoise"][0])
    assert np.abs(r["Wi"][g["rows"]] - g["Wi_rows"]).max() <= 1e-9 * np.abs(g["Wi_rows"]).max()
    assert np.abs(r["L"][g["rows"]] - g["L_rows"]).max() <= 1e-12


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_lean_large_n_oracle_matches_the_pinned_oracle(kind):
    """oracle/make_golden_baseline.py:lean_exact (one N x N buffer, dpotrf/dpotrs/dpotri in place) produced the
    N=32768 fixture that the reference's 8-temporary pdinv cannot produce in 62 GB: same results as gp_oracle."""
    from oracle.make_golden_baseline import lean_exact
    X, Y = O.synthetic(900, 8, seed=3, Dy=2)
    var, ls, noise = O.default_theta(8, False)
    a = lean_exact(kind, X, Y, var, ls, noise, block=256)
    b = O.parameters_changed(kind, X, Y, var, ls, False, noise)
    assert abs(a["lml"] - b["lml"]) <= 1e-13 * abs(b["lml"])
    assert np.abs(a["alpha"] - b["alpha"]).max() <= 1e-11 * np.abs(b["alpha"]).max()
    assert abs(a["dvar"][0] - b["dvar"]) <= 1e-10 * abs(b["dvar"])
    assert abs(a["dlen"][0] - b["dlen"][0]) <= 1e-10 * abs(b["dlen"][0])
    assert abs(a["dnoise"][0] - b["dL_dnoise"]) <= 1e-10 * abs(b["dL_dnoise"])
    assert np.abs(a["Wi_rows"] - b["Wi"][a["rows"]]).max() <= 1e-11 * np.abs(b["Wi"]).max()
    assert np.abs(a["L_rows"] - b["L"][a["rows"]]).max() <= 1e-13
    assert np.abs(a["diag_dL_dK"] - b["diag_dL_dK"]).max() <= 1e-11 * np.abs(b["diag_dL_dK"]).max()


def test_lean_large_n_oracle_with_the_production_blocking_matches_the_pinned_oracle():
    """the N=32768 fixture came from lean_exact with block=2048 and the 2 x 2 split, pinned before only at N=900 /
    block=256.  Here: the production block size (several 2048-row build blocks per matrix block, a ragged last one) and an
    uneven 2 x 2 split at N > 4096, against the reference-pinned oracle, RBF D=8 like configs[3]."""
    from oracle.make_golden_baseline import lean_exact
    N = 4608
    X, Y = O.synthetic(N, 8, seed=0)
    var, ls, noise = O.default_theta(8, False)
    b = O.parameters_changed("rbf", X, Y, var, ls, False, noise)
    for n1 in (None, 2560):                       # the default N // 2 split and an uneven one (blocks of 2560 / 2048 rows)
        a = lean_exact("rbf", X, Y, var, ls, noise, block=2048, n1=n1)
        assert abs(a["lml"] - b["lml"]) <= 1e-12 * abs(b["lml"])
        assert np.abs(a["alpha"] - b["alpha"]).max() <= 1e-10 * np.abs(b["alpha"]).max()
        assert abs(a["dvar"][0] - b["dvar"]) <= 1e-9 * abs(b["dvar"])
        assert abs(a["dlen"][0] - b["dlen"][0]) <= 1e-9 * abs(b["dlen"][0])
        assert abs(a["dnoise"][0] - b["dL_dnoise"]) <= 1e-9 * abs(b["dL_dnoise"])
        assert np.abs(a["Wi_rows"] - b["Wi"][a["rows"]]).max() <= 1e-10 * np.abs(b["Wi"]).max()
        assert np.abs(a["L_rows"] - b["L"][a["rows"]]).max() <= 1e-12
        assert np.abs(a["diag_dL_dK"] - b["diag_dL_dK"]).max() <= 1e-10 * np.abs(b["diag_dL_dK"]).max()


def test_baseline_fixtures_are_self_consistent():
    """Cheap identities on the stored full-size outputs (no N^3 work): dnoise = sum diag(dL_dK);
    diag(dL_dK)_i = 0.5 (alpha_i^2 - Wi_ii) on the sampled rows."""
    for name in ("baseline_c2_rbf_n4096_d8", "baseline_c3s_matern52_ard_n6144_d32",
                 "baseline_c3_matern52_ard_n16384_d32", "baseline_c4_rbf_n32768_d8"):
        g = baseline_golden(name)
        assert g["alpha"].shape == (g["N"], 1)
        assert abs(g["diag_dL_dK"].sum() - g["dnoise"][0]) <= 1e-9 * abs(g["dnoise"][0])
        rows = g["rows"]
        wii = g["Wi_rows"][np.arange(rows.size), rows]
        assert np.abs(0.5 * (g["alpha"][rows, 0] ** 2 - wii) - g["diag_dL_dK"][rows]).max() <= 1e-9 * np.abs(wii).max()

#!/usr/bin/env python3
"""GPU: the single-CU 128 x 128 Cholesky (chain_dev.h: generated potf2 + look-ahead factor) against numpy on small SPD
matrices, the pivot index reported for a matrix that is not positive definite, and the inverse through pdinv.
    python tools/diag_check.py"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gpy_amd import _lib as L  # noqa: E402


def spd(n, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.geomspace(1.0, cond, n)
    return (Q * w) @ Q.T


def main():
    worst = 0.0
    for n in (16, 24, 32, 100, 128, 129, 256, 300, 512, 1000):
        A = spd(n, n)
        Lg, info, _ = L.potrf(A)
        Lr = np.linalg.cholesky(A)
        e = np.abs(np.tril(Lg) - Lr).max() / np.abs(Lr).max()
        up = np.abs(np.triu(Lg, 1)).max() if n > 1 else 0.0
        Ai, L2, ld, info2, _ = L.pdinv(A)
        ei = np.abs(Ai - np.linalg.inv(A)).max() / np.abs(np.linalg.inv(A)).max()
        worst = max(worst, e, ei)
        print("N=%4d info %d  |L - chol| / |L| = %.2e   upper max %.1e   |Ainv - inv| rel %.2e  logdet err %.1e" % (
            n, info, e, up, ei, abs(ld - np.linalg.slogdet(A)[1])))
    for n, bad in ((128, 5), (128, 77), (300, 131), (300, 299), (512, 16)):
        A = spd(n, 7)
        A[bad, bad] = -1.0
        _, info, _ = L.potrf(A)
        print("N=%d, not PD at pivot %d: info %d %s" % (n, bad + 1, info, "OK" if info == bad + 1 else "MISMATCH"))
        worst = max(worst, 0.0 if info == bad + 1 else 1.0)
    print("RESULT", "PASS" if worst < 1e-11 else "FAIL", worst)


if __name__ == "__main__":
    main()

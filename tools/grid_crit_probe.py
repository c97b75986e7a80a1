#!/usr/bin/env python3
"""GPU (diagnostics build): what the critical path crit(k) of the block-cyclic grid mode costs per step when NOTHING else runs --
the bound on 8-GPU strong scaling of csrc/grid.hip.  MI355GP_GRID_DBG_CRIT (WRONG RESULTS by construction):
  0  the full evaluation
  1  every update (near / part1 / bulk / W) skipped: crit(k) alone, T = N / nb dependent steps
  2  ... and its broadcasts skipped (loopback: device copies): factor + inverse of the diagonal tile, panel GEMM, X-row GEMM, copies
  3  ... and only phase (a) left: 2-D copy out, potrf_device, stats, memset, trtri_device, 2-D copy back
Loopback transport (all logical ranks on one GPU): launch counts and dependences are those of the multi-GPU run, the broadcasts
are device-to-device copies instead of RCCL calls.

    python tools/grid_crit_probe.py [N=32768] [nb=512]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355GP_LIB", os.path.join(ROOT, "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L  # noqa: E402
from gpy_amd import grid as G  # noqa: E402
from gpy_amd.datasets import default_theta, synthetic  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    X, Y = synthetic(N, 8, seed=0)
    var, ls, noise = default_theta(8, False)
    th = L.theta_vec(var, ls, False, 8)
    T = -(-N // nb)
    os.environ["MI355GP_GRID_FORCE_GENERIC"] = "1"
    for Pr, Pc in ((1, 1), (2, 4)):
        row = []
        for mode in (0, 1, 2, 3):
            os.environ["MI355GP_GRID_DBG_CRIT"] = str(mode)
            g = G.GridContext.loopback(Pr, Pc, nb)
            try:
                g.set_data(X, Y)
                ms = []
                for _ in range(3):
                    _, r = g.exact_inference("rbf", False, th, noise, want_stage_ms=True)
                    ms.append(r["stage_ms"]["factor"])
                row.append(min(ms[1:]))
            finally:
                g.close()
        os.environ.pop("MI355GP_GRID_DBG_CRIT", None)
        print("N=%d nb=%d grid %dx%d (T = %d steps): factor stage %.1f ms | crit alone %.1f ms = %.3f ms per step | without its "
              "broadcasts %.1f = %.3f per step | phase (a) alone %.1f = %.3f per step" % (
                  N, nb, Pr, Pc, T, row[0], row[1], row[1] / T, row[2], row[2] / T, row[3], row[3] / T))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU (diagnostics build): the near-tile owners' path of every row of the persistent Cholesky, next to what the chain wanted.
Per row i, microseconds relative to the END of factor(i-2) (dcnt = i-1 is published ~3 us later): the LAST task of tiles
(i, i-2) [solve], (i, i-1) [last column + hand-over], (i, i) [last column + hand-over]: picked / computed / published; then
the chain's step i-1: factor ends, the solver waves have tile (i, i-1), their solve ends, the update starts (barrier Y: the
factor waves have tile (i, i)).  Late hand-overs show as `have P` > `factor ends` and `update starts` > `solve ends` + 1.

    python tools/persist_near.py 4096 [rows=all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355GP_LIB", os.path.join(ROOT, "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    r = L.dbg_persist(n, reps=5)
    st, near, nt = r["steps"], r["near"], r["nt"]
    print("N=%d persistent potrf %.3f ms" % (n, r["ms_persist"]))
    print(" row | (i,i-2) picked  done  publ | (i,i-1) picked  done  publ | (i,i) picked  done  publ | chain: factor(i-1) ends  has P  solved  update starts | lost: sub  dia")
    lost_sub = lost_dia = 0.0
    for i in range(2, nt):
        base = st[i - 2, 1]
        f = lambda v: "%6.1f" % (v - base) if v > 0 else "     -"
        c = st[i - 1]
        ls, ld = max(0.0, c[2] - c[1] - 1.0), max(0.0, c[3] - c[6] - 1.5)
        lost_sub += ls
        lost_dia += ld
        print(" %3d |        %s %s %s |        %s %s %s |      %s %s %s |        %s %s %s %s | %5.1f %5.1f" % (
            i, f(near[i, 2, 0]), f(near[i, 2, 1]), f(near[i, 2, 2]), f(near[i, 1, 0]), f(near[i, 1, 1]), f(near[i, 1, 2]),
            f(near[i, 0, 0]), f(near[i, 0, 1]), f(near[i, 0, 2]), f(c[1]), f(c[2]), f(c[6]), f(c[3]), ls, ld))
    print(" chain lost waiting for the sub-diagonal tile %.0f us, for the diagonal tile %.0f us" % (lost_sub, lost_dia))


if __name__ == "__main__":
    main()

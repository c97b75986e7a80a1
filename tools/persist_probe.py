#!/usr/bin/env python3
"""GPU: the persistent dataflow Cholesky (persist.hip) against the launch-per-step schedule -- time, bitwise equality of L and
the chain's per-step timeline.   python tools/persist_probe.py 1024,4096 [kcap]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gpy_amd import _lib as L  # noqa: E402


def main():
    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096").split(",")]
    kcaps = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
    for n in sizes:
        for kc in kcaps:
            r = L.dbg_persist(n, reps=5, kcap=kc)
            st = r["steps"]
            nt = r["nt"]
            fac = st[:, 1] - st[:, 0]
            wait = st[:-1, 2] - st[:-1, 1]                   # the two tiles of row j+1 published by their owners
            load = st[:-1, 4] - st[:-1, 2]                   # their sc1 loads + the write-through of L_jj, drained
            solve = st[:-1, 3] - st[:-1, 4]                  # L(j+1,j) = A(j+1,j) L_jj^-T, its stores, the Y image
            upd = st[:-1, 5] - st[:-1, 3]
            core = st[:-1, 6] - st[:-1, 4]                   # the MFMA chains of the solve alone
            umf = st[:-1, 7] - st[:-1, 3]                    # the MFMA loop of the update alone
            step = np.diff(st[:, 0])
            print("N=%d kcap=%d: steps %.3f ms, persistent %.3f ms, mismatches %d, info %d, abort %d" % (
                n, kc, r["ms_steps"], r["ms_persist"], r["mismatches"], r["info"], r["abort"]))
            if nt > 1:
                print("   per chain step (us): factor %.1f | wait %.1f (max %.1f) | load+store %.1f | solve %.1f | update %.1f | "
                      "step %.1f (max %.1f) | solve MFMAs %.1f, update MFMAs %.1f" % (fac.mean(), wait.mean(), wait.max(), load.mean(),
                                                                                      solve.mean(), upd.mean(), step.mean(), step.max(),
                                                                                      core.mean(), umf.mean()))
                nr = r["near"]
                j = min(nt - 3, max(2, nt // 2))
                T = st[j, 1]                                  # factor(j) ends
                print("   step j=%d, times relative to the end of factor(j): dcnt %+.1f | tile (j+2,j) picked %+.1f solved %+.1f "
                      "published %+.1f | chain cnt[j+1]: update starts %+.1f | tile (j+2,j+1) picked %+.1f computed %+.1f published "
                      "%+.1f | tile (j+2,j+2) picked %+.1f computed %+.1f published %+.1f | factor(j+1) starts %+.1f ends %+.1f, "
                      "solver has P at %+.1f" % ((j, st[j, 4] - T) + tuple(nr[j + 2, 2, :3] - T) + (st[j, 3] - T,) +
                                                 tuple(nr[j + 2, 1, :3] - T) + tuple(nr[j + 2, 0, :3] - T) +
                                                 (st[j + 1, 0] - T, st[j + 1, 1] - T, st[j + 1, 2] - T)))
                print("   chain publishes row j+1 at %+.1f (mean over steps: %.1f us after its update starts)" % (
                    nr[j + 1, 0, 3] - T, np.mean([nr[q + 1, 0, 3] - st[q, 3] for q in range(1, nt - 2)])))
                print("   wave 4: stores of L(j+1,j) issued %+.1f, drained %+.1f" % (nr[j + 1, 1, 3] - T, nr[j + 1, 2, 3] - T))
                if len(sys.argv) > 3:
                    for j in range(nt - 1):
                        print("   j=%2d %s" % (j, " ".join("%7.1f" % v for v in (st[j] - st[j, 0]))))


if __name__ == "__main__":
    main()

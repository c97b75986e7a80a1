import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gpy_amd import _lib as L
import os
for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096,2048").split(",")]:
    r = L.dbg_persist(n, reps=5)
    st = r["steps"]; nt = r["nt"]
    step = np.diff(st[:, 0]); fac = st[:, 1] - st[:, 0]; wait = st[:-1, 2] - st[:-1, 1]; solve = st[:-1, 3] - st[:-1, 4]; upd = st[:-1, 5] - st[:-1, 3]
    core = st[:-1, 6] - st[:-1, 4]
    print("N=%d persist %.3f ms" % (n, r["ms_persist"]))
    print(" step  :", " ".join("%5.1f" % v for v in step))
    print(" factor:", " ".join("%5.1f" % v for v in fac))
    print(" wait  :", " ".join("%5.1f" % v for v in wait))
    print(" solve :", " ".join("%5.1f" % v for v in solve))
    print(" core  :", " ".join("%5.1f" % v for v in core))
    print(" update:", " ".join("%5.1f" % v for v in upd))
    far = r["far"]
    print(" far tile (i, i-3), us relative to dcnt = i-2 (end of factor(i-3) + 3): last pass picked / done | solve picked / staged / solved / published")
    for i in range(4, nt, max(1, nt // 8)):
        base = st[i - 3, 1]
        f = far[i]
        print("   i=%2d: %s" % (i, " ".join("%7.1f" % (v - base) if v > 0 else "      -" for v in f[:6])))

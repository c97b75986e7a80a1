#!/bin/bash
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from gpy_amd import _lib as L
lib = L.lib()
for n in (8192, 16384, 8192, 16384):
    out = np.zeros(3)
    rc = lib.mi355gp_dbg_graph_factor(0, n, 5, out)
    print(n, "rc", rc, "launched %.3f ms  graph %.3f ms  nodes %d" % tuple(out), flush=True)
PY

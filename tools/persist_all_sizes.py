#!/usr/bin/env python3
"""GPU: the persistent Cholesky against the launch-per-step schedule at EVERY tile count it takes (nt = 2 .. 64, N = 128 nt, plus
ragged sizes), bitwise: mismatches / info / aborts must be 0 everywhere.  Prints one line per size that is not clean and a summary.

    python tools/persist_all_sizes.py [reps=2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpy_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bad = 0
sizes = [128 * nt for nt in range(2, 65)] + [300, 777, 1500, 2500, 3999, 5000, 7000]
for n in sizes:
    r = L.dbg_persist(n, reps=reps)
    if r["mismatches"] or r["info"] or r["abort"]:
        bad += 1
        print("N=%d: mismatches %d info %d abort %d" % (n, r["mismatches"], r["info"], r["abort"]))
print("%d sizes, %d not clean" % (len(sizes), bad))
sys.exit(1 if bad else 0)

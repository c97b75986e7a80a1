#!/usr/bin/env python3
"""GPU-box diagnostic: the (XCC, SE, SH, CU) coordinates workgroups land on, machine-wide and per CU-mask bit."""
import ctypes, collections, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpy_amd import _lib as L
lib = L.lib()
lib.mi355gp_dbg_cu_map.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
def run(nwg, bit):
    out = (ctypes.c_uint * (2 * nwg))()
    rc = lib.mi355gp_dbg_cu_map(0, nwg, bit, out)
    assert rc == 0, L.last_error()
    a = np.array(out, dtype=np.uint32).reshape(nwg, 2)
    hw, xcc = a[:, 0], a[:, 1] & 15
    return [(int(x), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 15), int((h >> 4) & 3), int(h & 15)) for h, x in zip(hw, xcc)]
allc = run(8192, -1)
cus = collections.Counter((x, se, sh, cu) for x, se, sh, cu, simd, w in allc)
print("distinct CUs seen:", len(cus))
byx = collections.defaultdict(set)
for (x, se, sh, cu) in cus: byx[x].add((se, sh, cu))
for x in sorted(byx): print("xcc", x, len(byx[x]), sorted(byx[x]))
print("first 24 workgroups land on:", [(x, se, sh, cu) for x, se, sh, cu, _, _ in allc[:24]])
for bit in list(range(0, 20)) + [32, 64, 128, 248, 255]:
    c = collections.Counter((x, se, sh, cu) for x, se, sh, cu, _, _ in run(64, bit))
    print("mask bit", bit, "->", dict(c))

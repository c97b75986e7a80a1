#!/usr/bin/env python
"""GPU: which hipEvent bracket makes the persistent launch miss its co-residency gate?  The timed steps of bench.py never see a
call-off, the two bracketed (profile) steps do about every second time.  200 evaluations at N=4096 per setting of the bracket
mask; prints the call-offs (each one costs 16 evaluations on launches, so ~11 is 'every eligible launch').
    python tools/bracket_calloff_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gpy_amd import _lib as L
from gpy_amd.datasets import default_theta, synthetic

N, D = 4096, 8
X, Y = synthetic(N, D, seed=0)
var, ls, noise = default_theta(D, False)
th = L.theta_vec(var, ls, False, D)
for fams in (None, ("potrf_persist",), ("trtri_early",), ("trtri", "lauum"), 1):
    c = L.Context(0)
    c.set_data(X, Y)
    for _ in range(6):
        c.exact_inference("rbf", False, th, noise, want_alpha=False)
    c.set_option("persist", 1)                      # calibration over: stay on the persistent schedule
    if fams is not None:
        c.set_option("profile", fams)
    a0 = c.get_option("persist_aborts")
    for _ in range(200):
        c.exact_inference("rbf", False, th, noise, want_alpha=False)
    print("brackets %-28s call-offs in 200 evaluations: %d" % (fams, c.get_option("persist_aborts") - a0), flush=True)
    c.close()

#!/usr/bin/env python3
"""GPU: where does the wall clock of one N = 32768 evaluation go that no stage accounts for (566 ms per step
against a 546 ms stage sum)?  Per call through the bare C-ABI: host wall time, the device-side stage sum, and the same with the
stage events off; then the drop-in classes' step.   python tools/c4_gap_probe.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpy_amd  # noqa: E402
from gpy_amd import _lib as L  # noqa: E402
from gpy_amd.datasets import default_theta, synthetic  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    D = 8
    X, Y = synthetic(N, D, seed=0)
    var, ls, noise = default_theta(D, False)
    th = L.theta_vec(var, ls, False, D)
    c = L.Context(0)
    c.set_data(X, Y)
    c.exact_inference("rbf", False, th, noise, want_alpha=False)
    for want in (True, False, True, False):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False, want_stage_ms=want)
            ts.append(1e3 * (time.perf_counter() - t0))
        extra = ""
        if want:
            st = r["stage_ms"]
            extra = "  stage sum %.1f (%s)" % (st["total"], " ".join("%s %.1f" % (k, v) for k, v in st.items() if k != "total"))
        print("N=%d C-ABI, stage events %s: wall %s ms%s" % (N, "on " if want else "off", " ".join("%.1f" % t for t in ts), extra))
    c.close()
    kern = gpy_amd.RBF(D, variance=var, lengthscale=ls)
    m = gpy_amd.GPRegression(X, Y, kern, noise_var=noise)
    x0 = m.param_array.copy()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        m.param_array = x0
        lml, g = m.log_likelihood(), m.gradient
        ts.append(1e3 * (time.perf_counter() - t0))
    print("N=%d drop-in classes: wall %s ms" % (N, " ".join("%.1f" % t for t in ts)))


if __name__ == "__main__":
    main()

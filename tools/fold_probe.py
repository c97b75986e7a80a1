#!/usr/bin/env python3
"""Timeline of the FOLDED persistent launch (csrc/persist.hip: potrf + L^-1 + X^T X as one tile dataflow) from its in-kernel
wall-clock stamps (mi355gp_dbg_fold): when each row of the inverse became final, when the chain ended, how the workers spent
their time.   python tools/fold_probe.py [N ...]   (MI355GP_PROBE_TUNES="0 1 2 3")"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpy_amd import _lib as L  # noqa: E402


def probe(N, tune, reps=3):
    out = np.zeros(8 + 2048)
    rc = L.lib().mi355gp_dbg_fold(0, N, reps, tune, out)
    if rc != 0:
        print("N=%d tune=%d: rc %d %s" % (N, tune, rc, L.last_error()))
        return
    nt = int(out[5])
    st = out[8:]
    t0 = st[0]
    us = lambda v: (v - t0) / 100.0          # noqa: E731  100 MHz ticks -> us
    xrow = [us(st[1 + i]) for i in range(nt)]
    chain_end = us(st[1 + nt])
    w = st[2 + nt:2 + nt + 4 * 255].reshape(-1, 4)
    w = w[w[:, 3] > 0]
    ends = (w[:, 3] - t0) / 100.0
    H = min(len(w) // 2, 3 * nt - 3)
    print("N=%d tune=%d: steps %.3f ms, folded %.3f ms; dX %.1e dW %.1e info %d" % (N, tune, out[0], out[1], out[2], out[3], out[4]))
    print("   chain end %.0f us; X rows final (us): %s" % (chain_end, " ".join("%.0f" % v for v in xrow[::max(1, nt // 16)])))
    print("   X last row %.0f us; workers end: min %.0f median %.0f max %.0f us" % (xrow[-1], ends.min(), np.median(ends), ends.max()))
    for name, sel in (("near", slice(0, H)), ("far", slice(H, None))):
        ww = w[sel]
        if len(ww):
            print("   %s owners (%d): mean busy us  P %.0f  X %.0f  W %.0f   (end %.0f)" % (
                name, len(ww), ww[:, 0].mean() / 100, ww[:, 1].mean() / 100, ww[:, 2].mean() / 100, ((ww[:, 3] - t0) / 100).mean()))


if __name__ == "__main__":
    Ns = [int(a) for a in sys.argv[1:]] or [4096]
    tunes = [int(v) for v in os.environ.get("MI355GP_PROBE_TUNES", "0 1 2 3").split()]
    for N in Ns:
        for tune in tunes:
            probe(N, tune)

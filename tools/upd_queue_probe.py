#!/usr/bin/env python3
"""GPU: the bounding experiment for a persistent trailing updater.

`MI355GP_DBG_UPD_QUEUE=1` makes the look-ahead Cholesky run every "part 2" update (the big K = 512 trailing updates: 30 of the
88 k_update_nt launches at N = 16384, 62 at N = 32768) from ONE resident launch of 512 workgroups that drains all their tiles from
a single atomic queue with every dependence ignored, while the panel stream runs diag / trsm / in-panel updates / part 1 as
always (minus its waits for part 2).  Same tile code, same operands, same traffic -- wrong numbers.  What it measures: the time
the real schedule loses to launch boundaries, per-launch ramp-up / drain and chain waits, i.e. an UPPER bound on what a dataflow
updater with per-panel ready flags could gain (a real one still has to wait for its panels).

    python tools/upd_queue_probe.py [N ...]        alternating off / on, three times each, mi355gp_bench_factor (device-only)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
os.environ.setdefault("MI355GP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L  # noqa: E402


def run(n, on):
    os.environ["MI355GP_DBG_UPD_QUEUE"] = "1" if on else "0"
    try:
        return L.bench_factor(n, reps=3)
    except L.MI355GPError:
        raise
    finally:
        os.environ.pop("MI355GP_DBG_UPD_QUEUE", None)


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [16384, 32768]
    for n in sizes:
        rows = {0: [], 1: []}
        for rep in range(3):
            for on in (0, 1):
                r = run(n, on)
                rows[on].append((r["potrf_ms"], r["trtri_ms"], r["lauum_ms"]))
        for on in (0, 1):
            p = [v[0] for v in rows[on]]
            t = [v[1] for v in rows[on]]
            print("N=%d part-2 queue %s: potrf %s ms (min %.2f)   trtri exposed %s (min %.2f)   potrf + trtri min %.2f" % (
                n, "ON " if on else "off", " ".join("%.2f" % v for v in p), min(p), " ".join("%.2f" % v for v in t), min(t),
                min(a + b for a, b in zip(p, t))))
        d = min(a + b for a, b, _ in rows[0]) - min(a + b for a, b, _ in rows[1])
        print("N=%d: bound on the gain of a dependence-free persistent part-2 updater: %.2f ms of potrf + exposed trtri" % (n, d))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU: what the STANDALONE factorisation (no inverse underneath: jitchol / pdinv's first stage / "Cholesky GF/s") could gain
from the schedule switches that exist.  Needs the diagnostics build (libmi355gp_diag.so: the switches are
compiled out of the product library).

    python tools/potrf_alone_probe.py [N ...]

Every variant: mi355gp_bench_factor with MI355GP_TRI_OVERLAP=0 (no early inverse), 3 repetitions, twice, alternating.
  base        the shipped schedule
  agg2        part 2 in pairs of panels (K = 1024 far updates: C read and written once per two panels)
  nbo1024     outer panels of 1024 columns
  p1main      part 1 on the main stream instead of the panel stream
  queue       BOUND, WRONG NUMBERS: every part-2 update from one resident launch with all dependences ignored
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355GP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L  # noqa: E402

VARIANTS = [
    ("base", {}),
    ("agg2", {"MI355GP_AGG2": "1"}),
    ("nbo1024", {"MI355GP_NBO": "1024"}),
    ("nbo256", {"MI355GP_NBO": "256"}),
    ("p1main", {"MI355GP_PART1_ON_PANEL": "0"}),
    ("queue", {"MI355GP_DBG_UPD_QUEUE": "1"}),
]


def run(n, env):
    keys = dict(env, MI355GP_TRI_OVERLAP="0")
    for k, v in keys.items():
        os.environ[k] = v
    try:
        return L.bench_factor(n, reps=3)["potrf_ms"]
    finally:
        for k in keys:
            os.environ.pop(k, None)


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [16384, 32768, 8192]
    for n in sizes:
        rows = {name: [] for name, _ in VARIANTS}
        for rep in range(2):
            for name, env in VARIANTS:
                rows[name].append(run(n, env))
        base = min(rows["base"])
        for name, _ in VARIANTS:
            v = rows[name]
            print("N=%d %-8s potrf alone %s ms  min %.2f  (%.1f TF/s, %+.1f %% vs base)" % (
                n, name, " ".join("%.2f" % x for x in v), min(v), n ** 3 / 3.0 / min(v) / 1e9, 100.0 * (min(v) / base - 1.0)))


if __name__ == "__main__":
    main()

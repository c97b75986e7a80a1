#!/usr/bin/env python3
"""GPU (diagnostics build): A/B of the MI355GP_PERSIST_TUNE bits of the persistent Cholesky on ONE box, alternating, factor only
(mi355gp_dbg_persist: persistent launch vs launch-per-step, bitwise comparison of L).

    python tools/persist_tune_ab.py 2048,3072,4096,4608 0,1,1024,1025 [rounds=3]
tune bits: 1 = far workers look at their tiles by ROW (the order up to round 5; default now: by column), 4 = no split hand-over,
8 = far workers do not reserve themselves for the tiles of the column the chain has reached, 16 = ... reserve one column earlier,
32 = near owners do not reserve themselves for the row the chain is about to reach, bits 16..18 = rows ahead they do (default 3),
64 / 128 = near ownership of 3 / 4 block diagonals, 1048576 / 2097152 = far tiles dealt out row- / column-major (default: rows up to nt = 32), bits 8..15 = share of near owners (workers / that number)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355GP_LIB", os.path.join(ROOT, "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L  # noqa: E402


def main():
    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2048,4096").split(",")]
    tunes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for n in sizes:
        res = {t: [] for t in tunes}
        bad = {t: 0 for t in tunes}
        for _ in range(rounds):
            for t in tunes:
                os.environ["MI355GP_PERSIST_TUNE"] = str(t)
                r = L.dbg_persist(n, reps=5)
                res[t].append(r["ms_persist"])
                bad[t] += r["mismatches"] + (1 if r["info"] or r["abort"] else 0)
        os.environ.pop("MI355GP_PERSIST_TUNE", None)
        for t in tunes:
            print("N=%d tune=%-6d persistent potrf %s ms  min %.3f  (mismatches / aborts: %d)" % (
                n, t, " ".join("%.3f" % v for v in res[t]), min(res[t]), bad[t]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU: the schedule a small factorisation ends up on (FactorWs::persist_auto: persistent launch + early inverse, or launches,
whichever the workspace's own timing says is faster on THIS box) and what each costs.   python tools/sched_probe.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpy_amd import _lib as L  # noqa: E402
from gpy_amd.datasets import default_theta, synthetic  # noqa: E402


def run(N, mode):
    D = 8
    X, Y = synthetic(N, D, seed=0)
    var, ls, noise = default_theta(D, False)
    th = L.theta_vec(var, ls, False, D)
    c = L.Context(0)
    c.set_data(X, Y)
    if mode != "auto":
        c.set_option("persist", 1 if mode == "persist" else 0)
    for _ in range(12):
        c.exact_inference("rbf", False, th, noise, want_alpha=False)
    t0 = time.perf_counter()
    for _ in range(200):
        info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False)
    dt = 1e3 * (time.perf_counter() - t0) / 200
    info, r = c.exact_inference("rbf", False, th, noise, want_alpha=False, want_stage_ms=True)
    st = r["stage_ms"]
    c.close()
    return dt, st, r["lml"]


def main():
    for N in [int(v) for v in sys.argv[1:]] or [2048, 3072, 4096]:
        out = {m: run(N, m) for m in ("persist", "steps", "auto")}
        print("N=%d  persistent %.3f ms (potrf %.3f)   launches %.3f ms (potrf %.3f)   auto %.3f ms (potrf %.3f)   lml equal: %s" % (
            N, out["persist"][0], out["persist"][1]["potrf"], out["steps"][0], out["steps"][1]["potrf"], out["auto"][0],
            out["auto"][1]["potrf"], out["persist"][2] == out["steps"][2] == out["auto"][2]))


if __name__ == "__main__":
    main()

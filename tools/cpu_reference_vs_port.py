#!/usr/bin/env python3
"""Where /root/reference exists: time ONE GP.parameters_changed of the REFERENCE'S OWN unmodified files (through
oracle/ref_loader.py, Cython extensions built by oracle/build_ref_cython.py into oracle/_ref) next to the port
(oracle/gp_oracle.py) on this machine's cores, same (X, Y, theta).  The paramz stand-in's Cache_this is a no-op, so the
reference recomputes K and r in the gradient step ("reference_uncached"); with the stand-in's memoising mode on
(oracle/paramz_stub/paramz/caching.py: K / _scaled_dist / dK_dr_via_X keyed like paramz's Cacher, limit 3) the gradient step
reuses the K and r of the inference step as a pip-installed GPy does ("reference_cached", the faithful number);
the port is timed both ways.
    python tools/cpu_reference_vs_port.py 4096,16384 profiles/r2_cpu_baseline_reference_vs_port.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref_cython, ref_loader  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def main():
    sizes = [int(v) for v in sys.argv[1].split(",")]
    out = sys.argv[2]
    build_ref_cython.build()
    ns = ref_loader.load()
    from threadpoolctl import threadpool_info
    threads = max(p.get("num_threads", 1) for p in threadpool_info())
    recs = []
    for kind, ARD, D in (("matern52", True, 32), ("rbf", False, 8)):
        for n in sizes:
            X, Y = O.synthetic(n, D, seed=0)
            var, ls, noise = O.default_theta(D, ARD)
            O.parameters_changed(kind, X[:512], Y[:512], var, ls, ARD, noise)
            t = {}
            t0 = time.perf_counter()
            r = ref_loader.run_iteration(ns, kind, X, Y, var, ls if ARD else float(ls[0]), ARD, noise)
            t["reference_uncached"] = time.perf_counter() - t0
            lml_ref = r["lml"]
            g_ref = [r["dvar"].copy(), r["dlen"].copy(), r["dnoise"].copy()]
            del r
            import paramz.caching as PC
            PC.ENABLED = True
            try:
                t0 = time.perf_counter()
                r = ref_loader.run_iteration(ns, kind, X, Y, var, ls if ARD else float(ls[0]), ARD, noise)
                t["reference_cached"] = time.perf_counter() - t0
            finally:
                PC.ENABLED = False
            assert r["lml"] == lml_ref and all((a == b).all() for a, b in zip(g_ref, [r["dvar"], r["dlen"], r["dnoise"]]))
            del r
            for cached in (True, False):
                t0 = time.perf_counter()
                p = O.parameters_changed(kind, X, Y, var, ls, ARD, noise, cached=cached)
                t["port_cached" if cached else "port_uncached"] = time.perf_counter() - t0
                assert abs(p["lml"] - lml_ref) <= 1e-10 * abs(lml_ref)
                del p
            rec = {"kind": kind, "ARD": ARD, "D": D, "N": n, "seconds": t, "blas_threads": threads, "host_cores": os.cpu_count(),
                   "cython": bool(ns.use_stationary_cython and ns.use_linalg_cython),
                   "what": "one GP.parameters_changed on the build container's CPU: the reference's own files (ref_loader, Cython "
                           "extensions built; uncached = the paramz stand-in recomputes K and r in the gradient step, cached = "
                           "K / _scaled_dist / dK_dr_via_X memoised as paramz's Cache_this(limit=3) does) vs the NumPy/SciPy port; "
                           "port_cached / reference_cached = %.3f" % (t["port_cached"] / t["reference_cached"])}
            print(json.dumps(rec), flush=True)
            recs.append(rec)
            json.dump({"records": recs}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

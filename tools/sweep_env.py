#!/usr/bin/env python3
"""GPU-box helper: time mi355gp_bench_factor / one full evaluation under different values of ONE environment switch,
each value in its own process (most switches are read once per process).
    python tools/sweep_env.py MI355GP_UPD64_MAX 0,128,192 --n 4096,8192 [--full]
Prints one line per (value, N): potrf / trtri / lauum ms, with --full also the whole evaluation and its LML bits."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys, struct
sys.path.insert(0, %r)
import numpy as np
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
os.environ.setdefault("MI355GP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L
from gpy_amd.datasets import synthetic, default_theta
out = {}
for n in %r:
    r = L.bench_factor(n, reps=%d)
    rec = {k: round(r[k], 4) for k in ("potrf_ms", "trtri_ms", "lauum_ms")}
    if %r:
        X, Y = synthetic(n, 8, seed=0)
        var, ls, noise = default_theta(8, False)
        c = L.Context(0); c.set_data(X, Y)
        th = L.theta_vec(var, ls, False, 8)
        c.exact_inference("rbf", False, th, noise)
        ms = []
        for _ in range(5):
            info, res = c.exact_inference("rbf", False, th, noise, want_stage_ms=True)
            ms.append(res["stage_ms"]["total"])
        rec["eval_ms"] = round(float(np.median(ms)), 4)
        rec["lml_hex"] = struct.pack(">d", res["lml"]).hex()
        rec["dtheta_hex"] = res["dtheta"].tobytes().hex()[:32]
        c.close()
    out[n] = rec
print("RESULT " + json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("values")
    ap.add_argument("--n", default="4096,8192")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    ns = [int(v) for v in a.n.split(",")]
    for val in a.values.split(","):
        env = dict(os.environ)
        if val == "unset":
            env.pop(a.var, None)
        else:
            env[a.var] = val
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ns, a.reps, a.full)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(a.var, val, "FAILED", r.stderr[-500:])
            continue
        res = json.loads(line[0][7:])
        for n in ns:
            print("%s=%s N=%d %s" % (a.var, val, n, json.dumps(res[str(n)])), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU-box comparator (NOT part of the product, which links no vendor BLAS): the same fp64 operations through PyTorch-ROCm,
i.e. rocBLAS / hipBLASLt (mm) and rocSOLVER / MAGMA (cholesky, cholesky_inverse), next to this library's device-resident
factorisation benchmark (mi355gp_bench_factor).  Prints one JSON line.
    python tools/vendor_compare.py [--n 16384]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, sync, reps):
    fn(); sync()
    ts = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--torch-only", action="store_true")
    a = ap.parse_args()
    n = a.n
    out = {"n": n}
    if not a.torch_only:
        # this library in a child process: PyTorch ships its own HIP runtime and the two do not share a device in one process
        import subprocess
        child = ("import sys, json; sys.path.insert(0, %r); from gpy_amd import _lib as L; bf = L.bench_factor(%d, reps=%d); "
                 "print('RESULT ' + json.dumps({k: round(bf[k], 3) for k in ('potrf_ms', 'trtri_ms', 'lauum_ms')}))"
                 % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, a.reps))
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        out["mi355gp"] = json.loads(line[0][7:]) if line else {"error": r.stderr[-300:]}
        if line:
            out["mi355gp"]["potri_ms"] = round(out["mi355gp"]["trtri_ms"] + out["mi355gp"]["lauum_ms"], 3)
    import torch
    dev = torch.device("cuda:0")
    sync = torch.cuda.synchronize
    g = torch.Generator(device=dev); g.manual_seed(0)
    B = torch.randn(n, n, dtype=torch.float64, device=dev, generator=g)
    C = torch.empty_like(B)
    ms = timed(lambda: torch.mm(B, B.t(), out=C), sync, a.reps)
    out["torch_mm_nt"] = {"ms": round(ms, 3), "tflops": round(2.0 * n ** 3 / (ms * 1e-3) / 1e12, 2)}
    A = C / n + torch.eye(n, dtype=torch.float64, device=dev) * 2.0          # SPD
    del B
    Lf = torch.empty_like(A)
    try:
        ms = timed(lambda: torch.linalg.cholesky(A, out=Lf), sync, a.reps)
        out["torch_cholesky"] = {"ms": round(ms, 3), "tflops": round(n ** 3 / 3.0 / (ms * 1e-3) / 1e12, 2)}
        ms = timed(lambda: torch.cholesky_inverse(Lf, out=C), sync, max(1, a.reps - 1))
        out["torch_cholesky_inverse"] = {"ms": round(ms, 3), "tflops": round(2.0 * n ** 3 / 3.0 / (ms * 1e-3) / 1e12, 2)}
    except Exception as e:                                            # backend missing on this build
        out["torch_error"] = repr(e)[:300]
    out["torch"] = torch.__version__
    try:
        out["torch_linalg_backend"] = str(torch.backends.cuda.preferred_linalg_library())
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Developer probe: the four operand layouts of the tiled fp64 MFMA GEMM at a few shapes (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpy_amd import _lib as L  # noqa: E402

shapes = [(4096, 4096, 512), (4096, 4096, 4096)] if len(sys.argv) < 2 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
rng = np.random.default_rng(0)
for (M, N, K) in shapes:
    for am, bn, name in ((0, 0, "NT"), (0, 1, "NN"), (1, 1, "TN"), (1, 0, "TT")):
        A = rng.standard_normal((K, M) if am else (M, K))
        B = rng.standard_normal((K, N) if bn else (N, K))
        C0 = np.zeros((M, N))
        _, ms = L.dbg_gemm(A, B, C0, am, bn, reps=10)
        mhz, cyc = L.dbg_gemm_clock()
        # workgroup 0: K/4 MFMA steps of 16 MFMAs per wave, 64 pipe cycles each, two waves per SIMD
        ideal = (K / 4.0) * 16 * 64
        print("gemm %s %dx%dx%d: %.3f ms  %.1f TF/s | wg0: %.0f MHz, %.0f cycles for the k-loop = %.2fx one wave's MFMA "
              "pipe time" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, mhz, cyc, cyc / ideal), flush=True)

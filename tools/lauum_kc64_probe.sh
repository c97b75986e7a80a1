#!/bin/bash
# X^T X below nt = 24 (64 x 64 quadrant items): the stage time against the chunk length (MI355GP_LAUUM_KC64, rows).
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
export TMPDIR=/tmp
for n in 1536 2048 2560 2944; do for kc in 256 512 768 1024 2048; do
MI355GP_LAUUM_KC64=$kc python - <<PY
from gpy_amd import _lib as L
r = [L.bench_factor($n, reps=5)["lauum_ms"] for _ in range(3)]
print("N=%d KC64=%-5s lauum %.4f %.4f %.4f ms" % ($n, "$kc", r[0], r[1], r[2]))
PY
done; done

#!/bin/bash
# X^T X of a small matrix from the work list with cut k ranges (128 x 128 items from nt = 24): the stage time against the chunk
# length, next to the length the host-side packing model picks (gemm.hip lauum_kc_rows_128), and against the single launch of
# whole-K tiles (MI355GP_LAUUM_SPLIT=0).   tools/lauum_kc_probe.sh
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
export MI355GP_LIB=${MI355GP_LIB:-$PWD/gpy_amd/libmi355gp_diag.so}
export TMPDIR=/tmp
one() {   # N KC(0 = the model's choice) SPLIT
    MI355GP_LAUUM_KC=$2 MI355GP_LAUUM_SPLIT=${3:-1} python - <<PY
from gpy_amd import _lib as L
r = [L.bench_factor($1, reps=5)["lauum_ms"] for _ in range(3)]
print("N=%d split=%s KC=%-5s lauum %.4f %.4f %.4f ms" % ($1, "${3:-1}", "$2" if $2 else "model", r[0], r[1], r[2]))
PY
}
for n in 4608 5120 6144 7168 8192; do one $n 0 0; one $n 0 128; one $n 1024 128; one $n 2048 128; one $n 3072 128; done

"""GPU: part-1 shaped updates (4 tile columns below the diagonal), K = 512, for MI355GP_UPD64_MAX = the environment's value"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the switches driven here exist only in the diagnostics build of the library (make -C gpy_amd/csrc diag)
os.environ.setdefault("MI355GP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpy_amd", "libmi355gp_diag.so"))
from gpy_amd import _lib as L
for ntr in (20, 36, 52, 68, 84, 100, 116, 128):
    ms = L.dbg_update_rect(ntr, 4, [512], 10)
    T = (ntr - 4) * 4
    print("rows %3d x 4 cols: %4d tiles  %.1f us  %.1f TF" % (ntr - 4, T, 1e3 * ms[0], T * 128 * 128 * 2 * 512 / ms[0] / 1e9))

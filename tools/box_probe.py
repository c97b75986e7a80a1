#!/usr/bin/env python3
"""GPU: what kind of box is this?  The same binary runs the persistent dataflow Cholesky at full or at HALF speed depending on
the box.  One line per call: the micro-benchmarks of mi355gp_dbg_peaks (fp64 MFMA rate, shader clock under
load, HBM copy rate), the persistent launch against the launch-per-step schedule at N = 2048 / 4096, and what rocm-smi says.
    python tools/box_probe.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpy_amd import _lib as L  # noqa: E402


def main():
    pk = L.dbg_peaks()
    p2, p4 = L.dbg_persist(2048, reps=5), L.dbg_persist(4096, reps=5)
    print("peaks: MFMA %.1f TF/s, shader %.0f MHz under load, HBM copy %.0f GB/s, fill %.0f GB/s | N=2048 steps %.3f persistent %.3f ms | "
          "N=4096 steps %.3f persistent %.3f ms (%s box)" % (pk["mfma_f64_tflops"], pk["shader_mhz_under_load"], pk["hbm_copy_gbs"],
                                                          pk["hbm_fill_gbs"], p2["ms_steps"], p2["ms_persist"], p4["ms_steps"],
                                                          p4["ms_persist"], "FAST" if p4["ms_persist"] < 2.2 else "SLOW"))
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmemuse", "--showtopo"], capture_output=True, text=True,
                             timeout=30).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "socclk", "Power", "NUMA", "Partition"))]
        print(" | ".join(keep)[:1500])
        out = subprocess.run(["rocm-smi", "--showcomputepartition", "--showmemorypartition"], capture_output=True, text=True, timeout=30).stdout
        print(" | ".join(ln.strip() for ln in out.splitlines() if "artition" in ln)[:600])
    except Exception as e:                                    # noqa: BLE001
        print("rocm-smi:", e)


if __name__ == "__main__":
    main()

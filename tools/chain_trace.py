#!/usr/bin/env python3
"""Where does a small factorisation spend its time?  From a rocprofv3 kernel trace of ONE workload, take the last evaluation
(after the last k_kbuild) and print, per kernel family, launches / summed duration / share of the span, the summed idle gaps
between consecutive kernels (no kernel running at all), and the first `--show` kernels with start offsets.
    python tools/chain_trace.py OUT/run_kernel_trace.csv [--show 40]"""
import csv
import sys


def short(n):
    for k in ("k_update_nt64", "k_update_nt", "k_diag128", "k_trsm128", "k_lauum64", "k_lauum", "k_trtri_stage64", "k_trtri_stage1_steal",
              "k_trtri_stage", "k_kbuild", "k_grad", "k_inv128", "k_trmv", "k_scalars", "k_reduce", "k_scale"):
        if k in n:
            return k
    return n[:24]


def main():
    path = sys.argv[1]
    show = int(sys.argv[sys.argv.index("--show") + 1]) if "--show" in sys.argv else 0
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"]),
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
    kb = [i for i, e in enumerate(ev) if e[2] == "k_kbuild"]
    s = kb[-1]
    seg = ev[s:]
    t0 = seg[0][0]
    span = (max(e[1] for e in seg) - t0) / 1e3
    print("%s: last evaluation spans %.1f us, %d kernels" % (path, span, len(seg)))
    fam = {}
    for e in seg:
        f = fam.setdefault(e[2], [0, 0.0])
        f[0] += 1
        f[1] += (e[1] - e[0]) / 1e3
    for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("  %-22s %4d launches %9.1f us  avg %7.1f" % (k, n, d, d / n))
    # idle time: union of busy intervals
    busy, cur_s, cur_e = 0.0, seg[0][0], seg[0][1]
    for e in seg[1:]:
        if e[0] > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = e[0], e[1]
        else:
            cur_e = max(cur_e, e[1])
    busy += cur_e - cur_s
    print("  no kernel running: %.1f us of %.1f" % (span - busy / 1e3, span))
    # potrf part: up to the last k_diag128 (+2 kernels)
    last_d = max(i for i, e in enumerate(seg) if e[2] == "k_diag128")
    p_end = seg[min(last_d + 2, len(seg) - 1)][1]
    print("  potrf span: %.1f us" % ((p_end - seg[1][0]) / 1e3))
    for e in seg[:show]:
        print("   +%8.1f  %7.1f us  q%-3d wg%-5d %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[4], e[2]))


if __name__ == "__main__":
    main()

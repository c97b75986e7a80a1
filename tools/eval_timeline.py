#!/usr/bin/env python3
"""Timeline of ONE evaluation from a rocprofv3 kernel trace: every kernel between two consecutive k_kbuild launches with its
start / end relative to the first, its queue and its grid -- where the exposed tails of a small evaluation (trtri after the
persistent Cholesky, X^T X, the solves on the side stream) really sit.

    rocprofv3 --output-format csv --kernel-trace -d OUT -o run -- python bench.py --n 4096 --d 8 --kind rbf --iso --steps 6 ...
    python tools/eval_timeline.py OUT/.../run_kernel_trace.csv [evaluation index from the end, default 2]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                 int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
    kb = [i for i, e in enumerate(ev) if e[2].startswith("k_kbuild")]
    if len(kb) < back + 1:
        print("not enough evaluations in the trace")
        return
    a, b = kb[-back - 1], kb[-back]
    t0 = ev[a][0]
    qs = sorted(set(e[3] for e in ev[a:b]))
    print("evaluation %d from the end: %d kernels, %.1f us from the start of k_kbuild to the next k_kbuild; queues %s" % (
        back, b - a, (ev[b][0] - t0) / 1e3, qs))
    print("%9s %9s %8s  q  %6s  kernel" % ("start us", "end us", "dur us", "wgs"))
    for e in ev[a:b]:
        print("%9.1f %9.1f %8.1f  %d  %6d  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, qs.index(e[3]), e[4], e[2][:60]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into small tracked files under profiles/:
   <tag>_kernel_stats.csv          rocprofv3 --stats table of the default bench.py workload
   <tag>_pmc_summary.csv           per kernel: dispatches, summed / average counter values (SQ pass, FETCH_SIZE, WRITE_SIZE)
   <tag>_traffic.json              per kernel HBM-side bytes per launch, corrected as MI355X_MICROARCH.md prescribes:
                                   FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B? -- rocprofv3 reports them in
                                   kilobytes; FETCH_SIZE counts 128-B requests as 64 B on gfx950 for wide coalesced reads,
                                   so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is left uncorrected (uncalibrated).
   <tag>_sparse_kernel_stats.csv   the same for bench.py --sparse (configs[4]); <tag>_c2_* for configs[1] (N=4096)
usage: python tools/summarize_profile.py gpurun_out/prof_r1 r1
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def agg_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not path:
        return agg
    for r in csv.DictReader(open(path)):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    for sub, name in (("stats", "%s_kernel_stats.csv" % tag), ("sparse_stats", "%s_sparse_kernel_stats.csv" % tag),
                      ("c2_stats", "%s_c2_kernel_stats.csv" % tag)):
        f = find(os.path.join(src, sub), "kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(out, name))
    for prefix, label in (("pmc", ""), ("sparse_pmc", "sparse_"), ("c2_pmc", "c2_")):
        merged = collections.defaultdict(dict)
        for p in ("a", "b", "c"):
            for k, cs in agg_counters(find(os.path.join(src, "%s_%s" % (prefix, p)), "counter_collection.csv")).items():
                merged[k].update(cs)
        if not merged:
            continue
        with open(os.path.join(out, "%s_%spmc_summary.csv" % (tag, label)), "w") as f:
            f.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
            for k in sorted(merged):
                for c in sorted(merged[k]):
                    n, s = merged[k][c]
                    f.write('"%s",%s,%d,%.6g,%.6g\n' % (k, c, n, s, s / n))
        traffic = {}
        for k, cs in merged.items():
            if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
                fn, fs = cs.get("FETCH_SIZE", [1, 0.0])
                wn, ws = cs.get("WRITE_SIZE", [1, 0.0])
                traffic[k] = {"launches_profiled": fn, "fetch_bytes_per_launch": 2.0 * 1024.0 * fs / max(fn, 1),
                              "write_bytes_per_launch": 1024.0 * ws / max(wn, 1)}
                traffic[k]["hbm_bytes_per_launch"] = (traffic[k]["fetch_bytes_per_launch"] +
                                                      traffic[k]["write_bytes_per_launch"])
        if not traffic:                                       # the FETCH_SIZE / WRITE_SIZE passes did not run or failed: no empty table
            continue                                          # (an empty one shadowed the previous round's numbers in round 4)
        with open(os.path.join(out, "%s_%straffic.json" % (tag, label)), "w") as f:
            json.dump({"note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md 'HBM'); "
                               "WRITE_SIZE uncorrected; units: bytes per launch, averaged over the profiled launches",
                       "kernels": traffic}, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()

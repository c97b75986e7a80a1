#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r2 > gpurun_out/prof_r2.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r2 r2 >> gpurun_out/prof_r2.log 2>&1
mkdir -p gpurun_out/prof_r2_summary; cp profiles/r2_* gpurun_out/prof_r2_summary/
python tools/potrf_timeline.py $(find gpurun_out/prof_r2/stats -name "*kernel_trace.csv" | head -1) > gpurun_out/prof_r2_summary/r2_potrf_timeline.txt 2>&1
find gpurun_out/prof_r2 -name "*.csv" -size +2M -delete; find gpurun_out/prof_r2 -name "*.db" -delete
tail -5 gpurun_out/prof_r2.log
O=gpurun_out/r2i; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 200 python bench.py --sparse --steps 10 --warmup 3 > $O/bench_sparse.json 2>> $O/bench.err
for n in 2048 4096 8192 32768; do timeout 300 python bench.py --n $n --d 8 --kind rbf --iso --steps 10 --warmup 3 --no-grid-leg --no-cpu-baseline > $O/bench_n$n.json 2>> $O/bench.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2i/bench*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d.get("stage_ms"), d.get("iteration_frac_of_fp64_peak"), d.get("roofline", {}).get("frac"), d.get("host_path"))
    except Exception as e: print(f, "ERR", e)
PY

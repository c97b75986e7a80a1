#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 --durations=6 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['stage_ms'], d['host_path'], d['grid']['ms_per_step'], d['cpu_baseline']['value'])"
tail -3 $O/bench.err
python tools/sweep_env.py MI355GP_TRI64_MAX unset --n 2048,4096,8192,16384 --reps 3 --full 2>&1 | tee $O/single.log
timeout 120 python bench.py --n 4096 --d 8 --kind rbf --iso --steps 50 --warmup 5 --no-grid-leg --no-cpu-baseline > $O/bench_c2.json 2>>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('C2', d['ms_per_step'], d['stage_ms'], d['host_path'], d['iteration_frac_of_fp64_peak'])"

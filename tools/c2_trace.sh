cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
MI355GP_LIB=$R/gpy_amd/libmi355gp_diag.so timeout 300 python $R/tools/persist_tune_ab.py 2048,3072,4096,4608 0,1048576 3 > $O/ab.txt 2>&1; cat $O/ab.txt
rm -rf /tmp/c2tr; timeout 300 rocprofv3 --output-format csv --kernel-trace -d /tmp/c2tr -o run -- python $R/bench.py --n 4096 --d 8 --kind rbf --iso --steps 40 --warmup 10 --no-legs --no-cpu-baseline --no-parity-gate > $O/c2_bench.json 2> $O/c2_prof.err
f=$(find /tmp/c2tr -name "run_kernel_trace.csv" | head -1); for b in 8 20; do python $R/tools/eval_timeline.py $f $b; done > $O/c2_timeline.txt
timeout 200 python $R/bench.py --n 4096 --d 8 --kind rbf --iso --steps 300 --warmup 20 --no-legs --no-cpu-baseline > $O/c2_plain.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/c2_plain.json').read().strip().splitlines()[-1]);print('C2 plain', d['ms_per_step'],d['stage_ms'])"

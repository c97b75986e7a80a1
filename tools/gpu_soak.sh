#!/bin/bash
# repeated full GPU suite + smoke on a fresh box: flakiness / box-dependence check
export TMPDIR=/tmp
O=gpurun_out/${1:-soak}; mkdir -p $O
for i in 1 2; do
( timeout 900 python -m pytest tests -m gpu -q --maxfail=5 ) > $O/pytest_$i.log 2>&1
grep -E "passed|failed|error" $O/pytest_$i.log | tail -2
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
rocm-smi --showuse --showmemuse 2>/dev/null | head -12

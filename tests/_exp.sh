python -m pytest tests/test_gpu_sparse.py -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error" | head
python bench.py --sparse --steps 5 --warmup 2 | cut -c1-120,600-1100

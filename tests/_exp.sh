python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --sparse --steps 5 --warmup 2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline

python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed"

python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | head

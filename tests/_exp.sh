python -m pytest tests/test_gpu_sum.py -m gpu -q -s 2>&1 | grep -a -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -15

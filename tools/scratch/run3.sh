python -m pytest tests/test_gpu_backward.py tests/test_gpu_autograd.py -q 2>&1 | grep -E "passed|failed"

python -m pytest tests/test_gpu_ops.py -q -k "pixelnorm or knobs" 2>&1 | tail -3

python -m pytest tests/test_gpu_mel.py -x -q 2>&1 | tail -3
python tools/fgla_bench.py 4 40 2>&1 | tail -1
python tools/fgla_bench.py 16 40 2>&1 | tail -1

for m in 0 1 2; do DDX_FGLA_FUSED=$m python tools/fgla_bench.py 4 40 2>&1 | tail -1; done
python -m pytest tests/test_gpu_mel.py -x -q 2>&1 | tail -3

python -m pytest tests/test_gpu_mel.py -x -q 2>&1 | tail -3
for m in 1 0; do DDX_FFT3=$m python tools/fgla_bench.py 4 40 2>&1 | tail -1; done
for m in 1 0; do DDX_FFT3=$m python tools/fgla_bench.py 16 40 2>&1 | tail -1; done

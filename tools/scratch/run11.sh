python -m pytest tests/test_gpu_ops.py -q -k "few_" 2>&1 | tail -3
python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_ddec.py tests/test_gpu_dae.py -x -q 2>&1 | tail -2
for b in 4 32; do
python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=$b new', j['ms_per_step'], j['roofline']['families_ms'])"
DDX_CONV_FEW=0 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=$b old', j['ms_per_step'], j['roofline']['families_ms'])"
done

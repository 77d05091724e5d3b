python -m pytest tests/test_gpu_ops.py -q -k "src0_alt or merged or conv_sm" 2>&1 | tail -2
python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -3
for b in 32 8 4; do
python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=$b new', j['ms_per_step'], j['roofline']['families_ms'])"
DDX_QKV_TWIN_MIN_PIXELS=0 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=$b old', j['ms_per_step'], j['roofline']['families_ms'])"
done

python -m pytest tests/test_gpu_ops.py -q -k "attention or merged" 2>&1 | tail -2
python -m pytest tests/test_gpu_unet.py tests/test_gpu_dae.py -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new ', j['ms_per_step'], j['roofline']['families_ms']['attention'])"
done
python bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=32 new ', j['ms_per_step'], j['roofline']['families_ms']['attention'])"

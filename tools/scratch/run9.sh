python -m pytest tests/test_gpu_ops.py -q -k "src0_alt or merged or conv_sm" 2>&1 | tail -3
python -m pytest tests/test_gpu_unet.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new ', j['ms_per_step'], j['roofline']['families_ms'])"
DDX_QKV_TWIN=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('old ', j['ms_per_step'], j['roofline']['families_ms'])"
done

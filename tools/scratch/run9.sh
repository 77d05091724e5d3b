for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new ', j['ms_per_step'], j['roofline']['families_ms'])"
DDX_DMA_FLAT=192 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('f192', j['ms_per_step'], j['roofline']['families_ms'])"
done

cd $GRAFT_REPO_ROOT
C=L0_skip_256_raw,L0_skip_512_raw,L0_skip_cat_raw,L1_skip_cat_raw,L0_skip_512
echo "== default"; timeout 300 python tools/conv_bench.py --path dma --cases $C 2>&1 | grep -v amdgpu
echo "== WIDE=0"; DDX_DMA_WIDE=0 timeout 300 python tools/conv_bench.py --path dma --cases $C 2>&1 | grep -v amdgpu
echo "== mfma"; timeout 300 python tools/conv_bench.py --path mfma --cases $C 2>&1 | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['roofline']['families_ms'])"
DDX_DMA_WIDE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide0', d['ms_per_step'], d['roofline']['families_ms'])"

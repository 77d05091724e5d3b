mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_ops.py -x -q -k "knobs or conv_dma or residual_up" 2>&1 | tail -3
Q=L0_skip_256_raw,L0_skip_cat_raw,L0_skip_cat2_raw,L1_skip_256_raw,L1_skip_512_raw,L1_skip_cat_raw,L1_skip_cat1_raw,L1_skip_cat2_raw,L2_skip_768_raw,L2_skip_cat_raw
python tools/conv_bench.py --cases $Q --path auto 2>&1 | grep -v amdgpu
echo SK64=0; DDX_DMA_SK64=0 python tools/conv_bench.py --cases $Q --path auto 2>&1 | grep -v amdgpu
echo forced dma 192; DDX_DMA_FLAT=192 python tools/conv_bench.py --cases L2_skip_768_raw,L2_skip_cat_raw,L2_skip_cat1_raw,L2_skip_cat2_raw,L2_skip_512_raw --path dma+mfma 2>&1 | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
DDX_DMA_SK64=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200

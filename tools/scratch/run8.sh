Q=L1_res1_raw,L0_up_res1_raw,L1_enc_res0_raw,L1_up_res1_raw,L1_dec1_res0_raw,L0_res1_enc_raw
python tools/conv_bench.py --cases $Q --epi real --path dma+dma16 2>&1 | grep -v amdgpu
echo SK32; DDX_DMA_SK32=1 DDX_DMA_WS=0 python tools/conv_bench.py --cases $Q --epi real --path dma 2>&1 | grep -v amdgpu
DDX_DMA_SK32=1 python tools/scratch/sk32_check.py 2>&1 | tail -1

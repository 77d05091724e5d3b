mkdir -p gpurun_out/r3
Q=Q0_skip_256_raw,Q0_skip_cat_raw,Q1_skip_512_raw,Q1_skip_cat_raw,L0_skip_256_raw,L0_skip_cat_raw,L0_skip_cat2_raw,L1_skip_512_raw,L1_skip_cat_raw,L2_skip_768_raw,L2_skip_cat_raw
python tools/conv_bench.py --cases $Q --path auto+mfma > gpurun_out/r3/q_one.log 2>&1
DDX_DMA_WIDE=0 python tools/conv_bench.py --cases $Q --path dma > gpurun_out/r3/q_one_narrow.log 2>&1
grep -v amdgpu gpurun_out/r3/q_one.log; grep -v amdgpu gpurun_out/r3/q_one_narrow.log

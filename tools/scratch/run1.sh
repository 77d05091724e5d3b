mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_ops.py -x -q -k "residual_up or conv_mfma or conv_dma or conv_sm or conv_direct" 2>&1 | tail -8
python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_ddec.py tests/test_gpu_dae.py -x -q 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
DDX_RES_UP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
python tools/conv_bench.py --cases L1_skip_512_raw,L0_up_skip_raw,L2_skip_768_raw,L1_up_skip_raw > gpurun_out/r3/one_up.log 2>&1; tail -4 gpurun_out/r3/one_up.log

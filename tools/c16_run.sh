cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py tests/test_gpu_ops.py tests/test_gpu_sampler.py tests/test_gpu_config3.py tests/test_gpu_dae.py tests/test_gpu_ddec.py -x -q > gpurun_out/c16_tests.log 2>&1; tail -5 gpurun_out/c16_tests.log

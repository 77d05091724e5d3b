#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_autograd.py tests/test_gpu_train_options.py tests/test_gpu_ddp_world2.py -x -q -m gpu 2>&1 | tail -8
for i in 1 2; do python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trtry_stats -o x --output-format csv -- python $root/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $out/trtry.log 2>&1
head -30 $out/trtry_stats/x_kernel_stats.csv | cut -c1-170 | grep -i "wpath\|Name"

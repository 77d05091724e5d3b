#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_autograd.py tests/test_gpu_train_options.py tests/test_gpu_ddp_world2.py -x -q -m gpu 2>&1 | tail -4
for k in 0 1 0 1; do echo "PER_BLOCK=$k $(DDX_WPATH_PER_BLOCK=$k python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["loss_mean"], d["grad_norm"])')"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trtry_stats -o x --output-format csv -- python $root/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $out/trtry.log 2>&1
head -40 $out/trtry_stats/x_kernel_stats.csv | cut -c1-170 | grep -i "wpath\|Name"

#!/bin/bash
# rocprofv3 kernel stats of the UNet training step (GPU box).  usage: tools/train_profile.sh <tag> [B]   -> gpurun_out/<tag>_train_kernel_stats.csv
tag=${1:-r01}; B=${2:-4}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_train_stats -o x --output-format csv -- python $root/tools/train_bench.py $B 3 > $out/${tag}_train_stats.log 2>&1
find $out/${tag}_train_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_train_kernel_stats.csv
tail -1 $out/${tag}_train_stats.log

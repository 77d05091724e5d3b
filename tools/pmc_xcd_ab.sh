#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the bench step under both conv_dma unit orders (GPU box).  Outputs under gpurun_out/xcd_ab/.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/xcd_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    DDX_DMA_XCD=$x rocprofv3 --pmc $c -d $out/x${x}_$c -o x --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/x${x}_$c.log 2>&1
    echo "== DDX_DMA_XCD=$x $c"; python $root/tools/pmc_summary.py $out/x${x}_$c conv_ | tee $out/x${x}_$c.txt
  done
done

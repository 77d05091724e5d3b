#!/bin/bash
# Sweep the tile configurations of the register-staged conv kernel (DDX_MFMA_FORCE) and the LDS-DMA kernel over conv_bench cases.
# usage: tools/conv_sweep.sh case1,case2,...     (GPU box)
cases=$1
echo -n "auto      : "; python tools/conv_bench.py --cases $cases --iters 40 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
echo -n "dma       : "; python tools/conv_bench.py --path dma --cases $cases --iters 40 2>&1 | grep -v amdgpu | awk '{printf "%s %s | ", $1, $3} END {print ""}'
for cfg in 128,32,1 128,64,1 128,32,2 128,64,2 128,32,4 128,64,4 256,32,1 256,64,1; do
  echo -n "$cfg: "; DDX_MFMA_FORCE=$cfg python tools/conv_bench.py --path mfma --cases $cases --iters 40 2>&1 | grep -v amdgpu | awk '{printf "%s %s | ", $1, $3} END {print ""}'
done

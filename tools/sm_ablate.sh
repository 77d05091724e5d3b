#!/bin/bash
# Timing ablations of the small-M conv kernels (conv_sm.hip, DDX_ABLATE bits: 1 no weight loads, 2 no operand DMA, 4 no MFMA,
# 8 no cross-wave reduction; wrong results, honest times).  Usage: tools/sm_ablate.sh [cases] [extra env ...]
CASES=${1:-L4_res1_raw,L4_res0_raw,L4_down_res0_raw,L4_proj_raw,L4_dec_skip_raw,L4_qkv_raw}
for d in 0 1 2 3 4 8 12 15; do
  echo "== DDX_ABLATE=$d"
  DDX_ABLATE=$d python tools/conv_bench.py --cases $CASES --path sm --epi real 2>&1 | grep -v amdgpu.ids
done

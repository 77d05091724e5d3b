#!/bin/bash
# L2 memory-side request counters per launch of single conv_dma layers (tools/conv_big_ab.py cases), by request size and as
# 32-byte DRAM units (FETCH_SIZE tallies every request at 64 B; see MI355X_MICROARCH.md).  Outputs under gpurun_out/pmc_case/.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_case
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r case; do
  i=$((i+1))
  p=0
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    p=$((p+1))
    rocprofv3 --pmc $c -d $out/c${i}_p$p -o x --output-format csv -- python $root/tools/conv_big_ab.py --iters 3 --cases "$case" > $out/c${i}_p$p.log 2>&1
    echo "== $case"; python $root/tools/pmc_summary.py $out/c${i}_p$p conv_dma | tee $out/c${i}_p$p.txt
  done
done <<CASES
${PMC_CASES:-L0 64->64 x8 plain
L0 64->64 x8
L0 32->64 x8 plain}
CASES

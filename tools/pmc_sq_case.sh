#!/bin/bash
# Wave-level counters of single conv_dma layers (tools/conv_big_ab.py cases): where the wave cycles go.  gpurun_out/pmc_sq/.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_sq
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r case; do
  i=$((i+1))
  p=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32" "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    p=$((p+1))
    rocprofv3 --pmc $c -d $out/c${i}_p$p -o x --output-format csv -- python $root/tools/conv_big_ab.py --iters 3 --cases "$case" > $out/c${i}_p$p.log 2>&1
    echo "== $case"; python $root/tools/pmc_summary.py $out/c${i}_p$p conv_dma | grep -v "^void\|^_Z" | tee $out/c${i}_p$p.txt
  done
done <<CASES
${PMC_CASES}
CASES

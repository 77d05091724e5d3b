#!/bin/bash
# Wave-level counters of the fused conv_res0 -> conv_res1 kernel (tools/pair_bench.py): where the wave cycles go.  gpurun_out/pmc_pair/.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_pair
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
p=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32" "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  p=$((p+1))
  rocprofv3 --pmc $c -d $out/p$p -o x --output-format csv -- python $root/tools/pair_bench.py ${1:-4} > $out/p$p.log 2>&1
  python $root/tools/pmc_summary.py $out/p$p conv_pair | tee $out/p$p.txt
done

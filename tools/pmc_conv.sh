#!/bin/bash
# PMC passes over one conv_bench configuration (GPU box).  usage: tools/pmc_conv.sh <outdir-under-gpurun_out> <conv_bench args...>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
            "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs -d $out/pass$i -o x --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_bench.py "$@" > $out/pass$i.log 2>&1
  echo "== pass $i"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pass$i conv_
done

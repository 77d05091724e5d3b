cd /tmp && export TMPDIR=/tmp
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES; do
rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/r3/fgla_pmc_$c -o x --output-format csv -- env DDX_FGLA_FUSED=0 python $GRAFT_REPO_ROOT/tools/fgla_bench.py 4 3 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES; do python tools/pmc_summary.py gpurun_out/r3/fgla_pmc_$c fgla; done

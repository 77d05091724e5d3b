#!/bin/bash
# rocprofv3 kernel stats (+ HBM-side PMC passes) of the secondary stages of the path on the GPU box: mel-STFT + FGLA, dual-window mel, MSS loss,
# VAE encode / decode, the sampling pipeline, the diffusion decoder and the training step.  usage: tools/profile_secondary.sh r02
# -> gpurun_out/<tag>_<stage>_kernel_stats.csv (+ _pmc.txt for the HBM-bound audio kernels) and <tag>_<stage>.log (the tool's own timing line)
tag=${1:-r02}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/${tag}_${name}_stats -o x --output-format csv -- "$@" > $out/${tag}_${name}.log 2>&1
  find $out/${tag}_${name}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_${name}_kernel_stats.csv
  tail -2 $out/${tag}_${name}.log
}
pmc() {     # name, kernel filter, command...
  local name=$1 filt=$2; shift 2
  : > $out/${tag}_${name}_pmc.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/${tag}_${name}_pmc_$c -o x --output-format csv -- "$@" > /dev/null 2>&1
    python $root/tools/pmc_summary.py $out/${tag}_${name}_pmc_$c "$filt" >> $out/${tag}_${name}_pmc.txt
  done
}
stats fgla python $root/tools/fgla_bench.py 4 20
pmc fgla "" python $root/tools/fgla_bench.py 4 4
stats msmel python $root/tools/msmel_bench.py 4
stats mss python $root/tools/mss_bench.py 2
pmc mss mss python $root/tools/mss_bench.py 2
stats vae python $root/tools/vae_profile.py
stats pipeline python $root/tools/pipeline_bench.py 4 20 20
stats ddec python $root/tools/ddec_bench.py 1
stats train python $root/tools/train_bench.py 8 3

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
# BASELINE configs[4] as quoted: B = 16, 100 steps (CFG + Heun), VAE decode, 200 FGLA iterations (the tool's own timing line only)
python $root/tools/pipeline_bench.py 16 100 200 > $out/${tag}_pipeline_b16.log 2>&1; tail -1 $out/${tag}_pipeline_b16.log
python $root/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_train_bench.json 2> $out/${tag}_train_bench.err; cat $out/${tag}_train_bench.json
# BASELINE configs[2] and configs[4] as bench.py lines (per-stage device times inside)
python $root/bench.py --mode config3 > $out/${tag}_config3.json 2> $out/${tag}_config3.err; cat $out/${tag}_config3.json
python $root/bench.py --mode sample > $out/${tag}_config5.json 2> $out/${tag}_config5.err; cat $out/${tag}_config5.json
# every tool's own timing line in one file
( for n in fgla msmel mss vae pipeline pipeline_b16 ddec train; do echo "== $n"; grep -v -E 'amdgpu.ids|rocprofv3|output_stream|simple_timer|^[WEI]20[0-9]{6} ' $out/${tag}_${n}.log | tail -6; done; echo "== graph purity (tools/graph_purity.py on the pipeline kernel trace)"; python $root/tools/graph_purity.py $out/${tag}_pipeline_stats/x_kernel_trace.csv; echo "== bench.py --mode train"; cat $out/${tag}_train_bench.json; echo "== bench.py --mode config3"; cat $out/${tag}_config3.json; echo "== bench.py --mode sample"; cat $out/${tag}_config5.json ) > $out/${tag}_secondary_timings.txt

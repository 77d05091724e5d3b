#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $out/trpmc_$c -o x --output-format csv -- python $root/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/trpmc_$c.log 2>&1
done
cd $root
python tools/pmc_summary.py $out/trpmc_FETCH_SIZE "" > $out/trpmc_summary.txt
python tools/pmc_summary.py $out/trpmc_WRITE_SIZE "" >> $out/trpmc_summary.txt
grep -A1 "wpath\|adamw\|sqnorm" $out/trpmc_summary.txt

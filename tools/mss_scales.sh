#!/bin/bash
# Per-scale kernel times of the MSS loss (GPU box): rocprofv3 kernel trace of tools/mss_bench.py, first half of the launches without
# gradient, second half with.  usage: tools/mss_scales.sh
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mssprof
rocprofv3 --kernel-trace --stats -d /tmp/mssprof -o x --output-format csv -- python $root/tools/mss_bench.py > /tmp/mss_bench.log 2>&1
grep mss_loss /tmp/mss_bench.log
f=$(find /tmp/mssprof -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('$f')):
    if 'mss' in r['Kernel_Name']:
        d[r['Kernel_Name'][:52]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    h = len(v) // 2
    print(f"{k:54s} {len(v):3d} launches: value only {sum(v[:h]) / h:7.0f} us, with gradient {sum(v[h:]) / (len(v) - h):7.0f} us")
PY

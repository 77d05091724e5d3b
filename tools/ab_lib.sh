#!/bin/bash
# A/B of two builds (variants/lib_base.so vs variants/lib_$1.so) on the conv cases, the decoder and the headline bench, ONE box
L=dualdiffusion_amd/lib/libddx_hip.so
cp variants/lib_base.so $L; python tools/conv_big_ab.py --save /tmp/ab.pt > /tmp/a.txt
cp variants/lib_$1.so $L; python tools/conv_big_ab.py --check /tmp/ab.pt > /tmp/b.txt
paste <(cut -c1-50 /tmp/a.txt) <(cut -c23-120 /tmp/b.txt)
for r in 1 2; do for v in base $1; do cp variants/lib_$v.so $L
  echo -n "$v: "; python tools/ddec_bench.py 1 | cut -c40-62
  echo -n "$v: "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done; done

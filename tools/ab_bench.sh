#!/bin/bash
# A/B of builds of libddx_hip.so on the headline bench inside ONE box: variants/lib_base.so vs variants/lib_<name>.so ...
L=dualdiffusion_amd/lib/libddx_hip.so
for r in 1 2; do
  for v in base "$@"; do
    cp variants/lib_$v.so $L
    echo -n "$v: "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done

#!/bin/bash
# A/B of an environment switch on the conv cases and the headline bench inside ONE box: tools/ab_env.sh VAR A B
for v in $2 $3; do echo "== $1=$v"; env $1=$v python tools/conv_big_ab.py ${AB_ARGS} | grep "L0\|L1\|ragged"; done
for r in 1 2; do for v in $2 $3; do echo -n "$1=$v: "; env $1=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done; done

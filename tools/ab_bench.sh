#!/bin/bash
# A/B of one run-time switch on ONE box (box-to-box spread is +-2-3 %, larger than most single changes):
#   tools/ab_bench.sh DDX_CONV_PAIR=0 [batch]      -> alternates the default build and the build with the switch, twice, prints ms per step + families
sw=${1:?switch, e.g. DDX_CONV_PAIR=0 (live switches: DDX_C16, DDX_CONV_PAIR, DDX_CONV_GEMM, DDX_CONV_DMA, DDX_SM_MAX_PIXELS, DDX_ABLATE bits)}; b=${2:-4}
for i in 1 2; do
  python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('default', j['ms_per_step'], j['roofline']['families_ms'])"
  env $sw python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$sw', j['ms_per_step'], j['roofline']['families_ms'])"
done

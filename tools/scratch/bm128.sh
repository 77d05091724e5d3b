#!/bin/bash
cd $GRAFT_REPO_ROOT
DDX_DMA_BM128=2 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -4
for k in 0 1 2 0 1 2; do
  echo "BM128=$k: $(DDX_DMA_BM128=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done

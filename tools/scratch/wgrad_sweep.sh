#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "512 48" "512 96" "512 192" "640 96" "512 48" "512 96" "512 192" "640 96"; do
  set -- $cfg
  echo "units=$1 cap=$2: $(DDX_WGRAD_UNITS=$1 DDX_WGRAD_CAP_MB=$2 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done

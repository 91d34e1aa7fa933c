#!/bin/bash
# Work items of 1 / 2 / 4 pixels (RT_ITEM_RAYS at 1024 samples) on the 4K scan frame and on rank 0's share of 4 and 8 ranks.
cd "$(dirname "$0")/.."
for W in 1 4 8; do for R in 1024 2048 4096; do
  RT_ITEM_RAYS=$R timeout -k 10 200 python bench.py --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$W item rays $R', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done; done

#!/bin/bash
# Work items of 2 / 4 pixels (RT_ITEM_RAYS at 1024 samples) and walks of 4 / 8 / 16 pixels on the 4K scan frame and on rank 0's share of 4 and 8 ranks.
cd "$(dirname "$0")/.."
for W in ${WORLDS:-1 4 8}; do for R in ${ITEM_RAYS:-2048 4096}; do for P in ${WALKS:-4 8 16}; do
  RT_ITEM_RAYS=$R RT_BEAM_WALK_PX=$P timeout -k 10 200 python bench.py --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$W item rays $R walk $P', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done; done; done

#!/bin/bash
# Walks of 4 / 8 / 16 pixels on a frame of DX x DY (default 1920 x 1080) and on rank 0's share of 4 and 8 ranks; items as the library sizes them.
cd "$(dirname "$0")/.."
for W in ${WORLDS:-1 4 8}; do for P in ${WALKS:-4 8 16}; do
  RT_BEAM_WALK_PX=$P timeout -k 10 200 python bench.py --dx ${DX:-1920} --dy ${DY:-1080} --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${DX:-1920}x${DY:-1080} W=$W walk $P', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done; done

#!/bin/bash
# What a launch of the beam kernel costs apart from its samples: the frame at 64, 256 and 1024 samples per pixel, whole and
# as the share of rank 0 of 8.  usage: fixed_cost.sh   (on the GPU box)
cd "$(dirname "$0")/.."
for W in 1 8; do for spp in 64 256 1024; do
  timeout -k 10 200 python bench.py --emulate-world $W --spp $spp --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$W spp=$spp', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done; done

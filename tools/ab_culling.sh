#!/bin/bash
# The libraries under raytrace_amd/variants side by side on the bench frame with the outline culling on in the timed steps.
cd "$(dirname "$0")/.."
for rep in 1 2; do for f in raytrace_amd/variants/*.so; do
  RT_LIB_PATH=$PWD/$f timeout -k 10 200 python bench.py --culling --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f culling', round(d['ms_per_step'],2), 'ms/step')"
done; done

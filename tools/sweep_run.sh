#!/bin/bash
# Runs bench.py once per library variant built by sweep_build.sh; prints value and kernel time.
cd "$(dirname "$0")/.."
for f in raytrace_amd/variants/*.so; do
  RT_LIB_PATH=$PWD/$f timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-culling-leg --no-cold-frame 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done

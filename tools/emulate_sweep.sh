#!/bin/bash
# Rank 0's share of a W-rank run of the bench on one GPU, W = 1 2 4 8 (scaling rehearsal); extra environment settings as arguments.
cd "$(dirname "$0")/.."
for W in ${WORLDS:-1 2 4 8}; do
  env "$@" timeout -k 10 200 python bench.py --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$W', '$*', round(d['ms_per_step'],2), 'ms/step; kernel', round(d['config']['rank0_kernel_ms_per_step'],2))" || exit 1
done

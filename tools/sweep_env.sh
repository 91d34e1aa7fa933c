#!/bin/bash
# usage: sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...   one bench run per environment setting
cd "$(dirname "$0")/.."
for spec in "$@"; do
  env $spec timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-culling-leg --no-cold-frame ${BENCH_ARGS} 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step')" || exit 1
done

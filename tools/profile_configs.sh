#!/bin/bash
# Kernel-trace statistics of tools/bench_configs.py for the named configurations.  Results: gpurun_out/cfg_<name>/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  OUT="$REPO/gpurun_out/cfg_$name"; mkdir -p "$OUT"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$REPO/tools/bench_configs.py" $name > "$OUT/run.log" 2>&1 || { echo "failed $name"; tail -5 "$OUT/run.log"; exit 1; }
  grep Mrays "$OUT/run.log"
  cut -d, -f1-4 "$OUT/run_kernel_stats.csv" | cut -c1-150 | head -8
done

#!/bin/bash
# Kernel-trace statistics of one default-shaped bench run.  Results: gpurun_out/stats/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/stats"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o bench --output-format csv -- python3 "$REPO/bench.py" "$@" > "$OUT/bench.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/bench.log"; exit 1; }
tail -1 "$OUT/bench.log" | cut -c1-400

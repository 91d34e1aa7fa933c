#!/bin/bash
# HBM traffic of a scene that does not fit the caches (tools/bench_configs.py scan_big: 21 M faces, Lambertian + light):
# one kernel-trace run for the durations, one counter pass each for FETCH_SIZE and WRITE_SIZE.  Results: gpurun_out/big/
# usage: profile_big.sh [config]     then: python tools/big_summary.py gpurun_out/big
REPO="$(cd "$(dirname "$0")/.." && pwd)"
name="${1:-scan_big}"
OUT="$REPO/gpurun_out/big"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$name" > "$OUT/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/stats.log"; exit 1; }
grep Mrays "$OUT/stats.log"
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $pass -d "$OUT/$pass" -o pmc --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$name" > "$OUT/$pass.log" 2>&1 || { echo "pass $pass failed"; tail -5 "$OUT/$pass.log"; exit 1; }
  echo "pass $pass done"
done

#!/bin/bash
# Kernel-trace statistics of tools/bench_orbit.py.  Results: gpurun_out/orbit/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/orbit"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$REPO/tools/bench_orbit.py" "$@" > "$OUT/run.log" 2>&1 || { echo failed; tail -5 "$OUT/run.log"; exit 1; }
grep charuco "$OUT/run.log"
python3 - "$OUT/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), "%9.2f ms total %9.1f us avg" % (float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY

#!/bin/bash
# Kernel-trace statistics of tools/bench_configs.py <config> for every library variant under raytrace_amd/variants.
# usage: sweep_profile.sh <config>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for f in "$REPO"/raytrace_amd/variants/librt_hip_*.so; do
  name=$(basename "$f" .so); OUT="$REPO/gpurun_out/sweep_$name"; mkdir -p "$OUT"
  RT_LIB_PATH="$f" timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$1" > "$OUT/run.log" 2>&1 || { echo "failed $name"; tail -5 "$OUT/run.log"; exit 1; }
  echo "== $name"; grep Mrays "$OUT/run.log" | cut -c1-50
  python3 - "$OUT/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(f'  {r["Name"][:44]:44s} {int(r["Calls"]):4d} calls {int(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
done

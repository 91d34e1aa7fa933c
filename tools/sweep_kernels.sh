#!/bin/bash
# Per-kernel GPU time of tools/bench_configs.py <config> for every library variant under raytrace_amd/variants.
# usage: sweep_kernels.sh <config> [kernel name filter ...]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cfg="$1"; shift
cd /tmp && export TMPDIR=/tmp
for f in "$REPO"/raytrace_amd/variants/librt_hip_*.so; do
  name=$(basename "$f" .so); OUT="$REPO/gpurun_out/sweepk_${cfg}_$name"; mkdir -p "$OUT"
  RT_LIB_PATH=$f timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$cfg" > "$OUT/run.log" 2>&1 || { echo "failed $name"; tail -5 "$OUT/run.log"; exit 1; }
  echo "== $name: $(grep Mrays "$OUT/run.log" | cut -c1-45)"
  python3 - "$OUT/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print("   %-34s %4s calls %9.3f ms" % (r["Name"].split("(")[0][-34:], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
done

#!/bin/bash
# Per-launch kernel durations of tools/bench_configs.py <config> for every library variant under raytrace_amd/variants.
# usage: sweep_launches.sh <config> [ENV=val ...]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cfg="$1"; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for f in "$REPO"/raytrace_amd/variants/librt_hip_*.so; do
  name=$(basename "$f" .so); OUT="$REPO/gpurun_out/launches_${cfg}_$name"; rm -rf "$OUT"; mkdir -p "$OUT"
  RT_LIB_PATH=$f timeout -s KILL 200 rocprofv3 --kernel-trace -d "$OUT" -o run --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$cfg" > "$OUT/run.log" 2>&1 || { echo "failed $name"; tail -5 "$OUT/run.log"; exit 1; }
  echo "== $name $*: $(grep Mrays "$OUT/run.log" | cut -c1-45)"
  python3 "$REPO/tools/launch_list.py" "$OUT"
done

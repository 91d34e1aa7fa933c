#!/bin/bash
# Runs tools/bench_configs.py with every library variant under raytrace_amd/variants (see sweep_build.sh).
# usage: sweep_configs.sh <config> ...
cd "$(dirname "$0")/.."
for f in raytrace_amd/variants/librt_hip_*.so; do
  echo "== $f"
  RT_LIB_PATH=$PWD/$f timeout -k 10 200 python -u tools/bench_configs.py "$@" 2>/dev/null | cut -c1-60 || exit 1
done

#!/bin/bash
# tools/bench_configs.py <configs...> once per library variant under raytrace_amd/variants.
# usage: sweep_cfg_variants.sh <config> [config ...]
cd "$(dirname "$0")/.."
for f in raytrace_amd/variants/librt_hip_*.so; do
  echo "== $(basename $f .so)"
  RT_LIB_PATH=$PWD/$f timeout -k 10 200 python -u tools/bench_configs.py "$@" 2>&1 | grep Mrays | cut -c1-110 || exit 1
done

#!/bin/bash
# When the persistent waves of k_beam start and end (diagnostics build raytrace_amd/variants/librt_hip_times.so, -DRT_WAVE_TIMES).
cd "$(dirname "$0")/.."
for W in ${WORLDS:-1 8}; do
  echo "== W=$W $*"
  env "$@" RT_DEBUG_BEAM=1 RT_LIB_PATH=$PWD/raytrace_amd/variants/librt_hip_times.so timeout -k 10 200 python bench.py --emulate-world $W --steps 2 --warmup 1 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>&1 | grep "k_beam waves" | tail -2 | cut -c1-700
done

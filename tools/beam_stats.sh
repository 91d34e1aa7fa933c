#!/bin/bash
# What k_beam's pixels are made of (diagnostics build raytrace_amd/variants/librt_hip_stats.so: tools/sweep_build.sh "stats -DRT_VISIT_STATS"):
# pixels that list faces, those with lists beyond one instalment or more survivors than the compact array holds, and their share of the wave-time.
cd "$(dirname "$0")/.."
for W in 1 8; do
echo "== W=$W"
RT_DEBUG_BEAM=1 RT_LIB_PATH=$PWD/raytrace_amd/variants/librt_hip_stats.so timeout -k 10 300 python bench.py --emulate-world $W --steps 1 --warmup 1 --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check 2>&1 | grep "pixel beams\|wave rounds" | tail -2
done

#!/bin/bash
# Builds librt_hip variants (waves per SIMD x LDS stack pairs) side by side for an A/B run on the GPU box.
set -e
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p ../variants
for w in 3 4 5 6 8; do
  for p in 8 12; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
      -DRT_FAST_WAVES=$w -DRT_LDS_PAIRS=$p -o ../variants/librt_hip_w${w}_p${p}.so rt_api.cpp render.cpp kernels.hip bvh_build.cpp &
  done
  wait
done
ls -la ../variants

#!/bin/bash
# Builds librt_hip variants side by side for an A/B run on the GPU box.
# usage: sweep_build.sh "<name> <extra hipcc flags>" ...
set -e
cd "$(dirname "$0")/../raytrace_amd/csrc"
rm -rf ../variants && mkdir -p ../variants
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
      "$@" -o ../variants/librt_hip_${name}.so rt_api.cpp render.cpp kernels.hip bvh_build.cpp obj_scan.cpp scene_json.cpp &
done
wait
ls ../variants

#!/bin/bash
# Vector instructions of a k_beam launch apart from its samples: SQ_INSTS_VALU / SQ_INSTS_SALU per launch at 512 and at 1024 samples per
# pixel, for every library under raytrace_amd/variants (knock-out builds tell the parts apart) or the tree's own.
cd "$(dirname "$0")/.."
libs=$(ls raytrace_amd/variants/*.so 2>/dev/null); [ -z "$libs" ] && libs=raytrace_amd/librt_hip.so
for f in $libs; do for spp in 512 1024; do
  rm -rf gpurun_out/pmc
  RT_LIB_PATH=$PWD/$f PMC_ARGS="--spp $spp --no-cpu-baseline --no-culling-leg --no-cold-frame --no-check --steps 2 --warmup 1" bash tools/profile_pmc.sh sq1 > /dev/null || exit 1
  echo "== $f spp $spp"; python tools/pmc_summary.py gpurun_out/pmc "k_beam<" | grep -E "INSTS_VALU|INSTS_SALU|INSTS_VMEM" | awk '{print $1, $5}'
done; done

"""MI355X-native implementation of raytrace's path-tracing loop behind a C ABI.

`raytrace_amd.phys` mirrors the reference's scene API; the pixels come from
raytrace_amd/librt_hip.so (HIP kernels for gfx950, sources in raytrace_amd/csrc).
"""
from . import abi, phys  # noqa: F401

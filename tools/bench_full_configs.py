#!/usr/bin/env python3
"""The teapot and material configurations at their full sizes (teapot 1920x1080 @ 256 spp, material 2048x2048 @ 4096 spp), fast mode,
one GPU: whole-frame time and rate.  usage: bench_full_configs.py [teapot] [material]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytrace_amd import phys, scenes

CONFIGS = {
    "teapot": lambda: scenes.teapot(dx=1920, dy=1080, spp=256),
    "material": lambda: scenes.material(dx=2048, dy=2048, spp=4096),
}
for name in sys.argv[1:] or list(CONFIGS):
    sc = CONFIGS[name]()
    with phys.DeviceScene(sc) as dev:
        t0 = time.perf_counter()
        art = dev.render(sc.Camera[0], sc.RenderOptions)
        dt = time.perf_counter() - t0
    st = art.Stats
    print(f"{name:9s} {sc.RenderOptions.Dx}x{sc.RenderOptions.Dy}@{sc.RenderOptions.RaysPerPixel}: {dt:.2f} s, {st.TotalRays/dt/1e6:.0f} Mrays/s "
          f"({st.TotalRays/1e9:.2f} G rays + {st.ShadowRays/1e9:.2f} G shadow rays, {st.TraverseLaunches} traversal launches), "
          f"mean byte {art.Image[..., :3].mean():.2f}", flush=True)

#!/usr/bin/env python3
"""Throughput of the other benchmark configurations (stochastic materials: shade / light-loop / scattered-ray
generations), fast mode, one GPU.  usage: bench_configs.py [name ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytrace_amd import abi, phys, scenes

CONFIGS = {
    "cornell": lambda: scenes.cornell(dx=256, dy=256, spp=64),
    "teapot": lambda: scenes.teapot(dx=1920, dy=1080, spp=32),
    "material": lambda: scenes.material(dx=1024, dy=1024, spp=32),
    "scan_lit": lambda: scenes.scan(level=7, dx=1920, dy=1080, spp=16, tex_size=1024, lit=True),
    # 21 M faces: 2.5 GB of float64 vertices + 1 GB of float32 faces + 1.3 GB of nodes, far beyond L2 + the 256 MB memory-side
    # cache; Lambertian + a point light, so every generation after the first is incoherent
    "scan_big": lambda: scenes.scan(level=10, dx=1920, dy=1080, spp=16, tex_size=1024, lit=True),
    "scan_big_emit": lambda: scenes.scan(level=10, dx=3840, dy=2160, spp=64, tex_size=4096),
}
reps = int(os.environ.get("BENCH_REPS", "1"))  # the fastest of that many timed frames
names = sys.argv[1:] or [n for n in CONFIGS if not n.startswith('scan_big')]
for name in names:
    sc = CONFIGS[name]()
    with phys.DeviceScene(sc) as dev:
        dev.render(sc.Camera[0], sc.RenderOptions)  # warm-up: queues allocated
        dt = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            art = dev.render(sc.Camera[0], sc.RenderOptions)
            dt = min(dt, time.perf_counter() - t0)
    st = art.Stats
    print(f"{name:9s} {sc.RenderOptions.Dx}x{sc.RenderOptions.Dy}@{sc.RenderOptions.RaysPerPixel}: {st.TotalRays/dt/1e6:8.1f} Mrays/s "
          f"({st.TotalRays/1e6:.1f} M rays + {st.ShadowRays/1e6:.1f} M shadow rays in {dt*1e3:.1f} ms; traversal launches {st.TraverseLaunches}, "
          f"{st.TraverseSeconds*1e3:.1f} ms inside them)", flush=True)

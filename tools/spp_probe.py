import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytrace_amd import abi, phys, scenes
name = sys.argv[1]
for spp in (4, 8, 16, 32):
    sc = scenes.teapot(dx=1920, dy=1080, spp=spp) if name == "teapot" else scenes.scan(level=7, dx=1920, dy=1080, spp=spp, tex_size=1024, lit=True)
    with phys.DeviceScene(sc) as dev:
        dev.render(sc.Camera[0], sc.RenderOptions)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); art = dev.render(sc.Camera[0], sc.RenderOptions); best = min(best, time.perf_counter() - t0)
    print(f"{name} spp {spp}: {best*1e3:.2f} ms, {art.Stats.TotalRays/best/1e6:.0f} Mrays/s, {best*1e3/spp:.3f} ms per sample", flush=True)

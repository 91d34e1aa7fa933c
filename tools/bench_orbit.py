#!/usr/bin/env python3
"""Frame loop of example/charuco (main.go:259-339): the scene stays resident, only the camera changes per frame.
Reports frames/s and Mrays/s over an orbit of the calibrated camera three ways: a host loop of rt_render, the
library's own loop (rt_render_frames) without overlap, and with copy-out overlapped with the next frame's trace."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytrace_amd import abi, phys, scenes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
leds = len(sys.argv) > 2 and sys.argv[2] == "leds"
base = scenes.charuco(dx=1440, dy=1080, spp=16, leds=leds)
cams = [scenes.charuco(dx=8, dy=8, spp=1, frame=f, frames_per_orbit=frames).Camera[0] for f in range(frames)]
o = base.RenderOptions
what = f"charuco{' + 257 lights' if leds else ''}: {frames} frames 1440x1080@16"
with phys.DeviceScene(base) as dev:
    dev.render(cams[0], o)
    out = np.zeros((frames, o.Dy, o.Dx, 4), dtype=np.uint8)
    out[:] = 1  # touch the pages: first-touch faults are the host's cost, not the loop's
    dev.render_frames(cams[:2], o, out=out[:2])
    for label, run in [("host loop of rt_render", None), ("rt_render_frames, no overlap", False), ("rt_render_frames, overlapped", True)]:
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            if run is None:
                rays = 0
                for cam in cams:
                    rays += dev.render(cam, o).Stats.TotalRays
            else:
                _, st = dev.render_frames(cams, o, overlap=run, out=out)
                rays = sum(s.TotalRays for s in st)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"{what}, {label}: {best:.3f} s = {frames/best:.1f} frames/s, {rays/best/1e6:.0f} Mrays/s, {best/frames*1e3:.2f} ms/frame")

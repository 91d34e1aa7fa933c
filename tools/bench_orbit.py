#!/usr/bin/env python3
"""Frame loop of example/charuco (main.go:259-339): the scene stays resident, only the camera changes per frame.
Reports frames/s and Mrays/s over an orbit of the calibrated camera."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytrace_amd import abi, phys, scenes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
leds = len(sys.argv) > 2 and sys.argv[2] == "leds"
base = scenes.charuco(dx=1440, dy=1080, spp=16, leds=leds)
cams = [scenes.charuco(dx=8, dy=8, spp=1, frame=f, frames_per_orbit=frames).Camera[0] for f in range(frames)]
with phys.DeviceScene(base) as dev:
    dev.render(cams[0], base.RenderOptions)
    t0 = time.perf_counter()
    rays = 0
    for cam in cams:
        art = dev.render(cam, base.RenderOptions)
        rays += art.Stats.TotalRays
    dt = time.perf_counter() - t0
print(f"charuco{' + 257 lights' if leds else ''}: {frames} frames 1440x1080@16 in {dt:.3f} s = {frames/dt:.1f} frames/s, {rays/dt/1e6:.0f} Mrays/s, "
      f"{dt/frames*1e3:.2f} ms/frame")

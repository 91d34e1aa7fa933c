#!/usr/bin/env python3
"""The 4K scan frame under a pinhole camera (and a focusable camera without aperture): pixel beams from one point against the
packet walk (RT_NO_BEAM=1).  usage: bench_pinhole.py [spp] [level]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytrace_amd import abi, phys, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dx, dy = 3840, 2160
base = scenes.scan(level=level, dx=dx, dy=dy, spp=spp, tex_size=4096)


def pinhole(origin, look_at, vup, fov_deg, aspect):
    o, a, up = (np.asarray(v, dtype=float) for v in (origin, look_at, vup))
    w = (o - a) / np.linalg.norm(o - a)
    u = np.cross(up, w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    hh = np.tan(np.radians(fov_deg) / 2); hw = hh * aspect
    return phys.PinholeCamera(tuple(o - hw * u - hh * v - w), tuple(o), tuple(2 * hw * u), tuple(2 * hh * v))


cams = {"pinhole": pinhole((300.0, 60.0, 120.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 45.0, dx / dy),
        "focusable f/inf": phys.FocusableCamera(LookFrom=(300.0, 60.0, 120.0), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0), FOVHeight=0.8,
                                                FOVWidth=0.8 * dx / dy, Aperture=0.0, WorkingDistance=330.0)}
for name, cam in cams.items():
    sc = phys.Scene(Camera=[cam], Node=base.Node, RenderOptions=phys.RenderOptions(0, spp, 10, dx, dy))
    for no_beam in ("", "1"):
        if no_beam:
            os.environ["RT_NO_BEAM"] = "1"
        else:
            os.environ.pop("RT_NO_BEAM", None)
        with phys.DeviceScene(sc) as dev:
            dev.render(cam, sc.RenderOptions, abi.MODE_FAST)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                art = dev.render(cam, sc.RenderOptions, abi.MODE_FAST)
                best = min(best, time.perf_counter() - t0)
        print(f"{name:16s} {'packet walk' if no_beam else 'pixel beams':11s}: {best * 1e3:7.1f} ms per {dx}x{dy}@{spp} frame, {art.Stats.TotalRays / best / 1e9:6.1f} Grays/s "
              f"(hit fraction {1 - art.Stats.RaysLeftScene / art.Stats.TotalRays:.2f}, culled {art.Stats.CulledRays / art.Stats.TotalRays:.2f})", flush=True)
os.environ.pop("RT_NO_BEAM", None)

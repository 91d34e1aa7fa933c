#!/usr/bin/env python3
"""Fast-mode traversal work per primary ray of the 4K scan frame (boxes = 2 per node visited)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytrace_amd import abi, phys, scenes
from oracle import binding as ob
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.scan(level=lvl, dx=3840, dy=2160, spp=1, tex_size=64)
ys, xs = np.mgrid[0:2160:4, 0:3840:4]
st = np.stack([(xs.ravel() + 0.5) / 3840, (ys.ravel() + 0.5) / 2160], 1)
o, d = ob.cast(sc.Camera[0], st, np.zeros((len(st), 2), dtype=np.uint32))
with phys.DeviceScene(sc) as dev:
    f = dev.intersect(o, d, 1e-6, 1.7976931348623157e308, abi.MODE_FAST)
    r = dev.intersect(o, d, 1e-6, 1.7976931348623157e308, abi.MODE_REFERENCE)
hit = f["node"] >= 0
print("rays", len(o), "hit frac", hit.mean())
for name, h in (("fast", f), ("reference", r)):
    print(name, "nodes-or-boxes/ray all %.1f hit %.1f miss %.1f | prims/ray all %.2f hit %.2f miss %.2f | max boxes %d" % (
        h["boxes_tested"].mean(), h["boxes_tested"][hit].mean(), h["boxes_tested"][~hit].mean(),
        h["prims_tested"].mean(), h["prims_tested"][hit].mean(), h["prims_tested"][~hit].mean(), h["boxes_tested"].max()))
print("agree", (f["prim"][hit] == r["prim"][hit]).mean(), np.array_equal(f["t"][hit], r["t"][hit]))

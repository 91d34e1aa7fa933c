#!/usr/bin/env python3
"""SIMD efficiency of the 8x8-tile-per-wave mapping on the 4K scan frame: per tile, the work of the
slowest lane against the mean lane (work = fast-mode node visits + 2 x face tests per primary ray)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytrace_amd import abi, phys, scenes
from oracle import binding as ob
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.scan(level=lvl, dx=3840, dy=2160, spp=1, tex_size=64)
tot_max = tot_mean = 0.0
tile_cost = []
with phys.DeviceScene(sc) as dev:
    for y0 in range(0, 2160, 240):
        rows = 240 if y0 + 240 <= 2160 else 2160 - y0
        ys, xs = np.mgrid[y0:y0 + rows, 0:3840]
        st = np.stack([(xs.ravel() + 0.5) / 3840, (ys.ravel() + 0.5) / 2160], 1)
        o, d = ob.cast(sc.Camera[0], st, np.zeros((len(st), 2), dtype=np.uint32))
        f = dev.intersect(o, d, 1e-6, 1.7976931348623157e308, abi.MODE_FAST)
        w = (f["boxes_tested"] / 2 + 2 * f["prims_tested"]).reshape(rows // 8, 8, 480, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        tot_max += w.max(1).sum(); tot_mean += w.mean(1).sum()
        tile_cost.append(w.max(1))
tile_cost = np.concatenate(tile_cost)
print("tiles", len(tile_cost), "lane efficiency (mean/max) %.3f" % (tot_mean / tot_max))
print("tile cost: mean %.1f  p50 %.1f  p99 %.1f  max %.1f  (max/mean %.1f)" % (
    tile_cost.mean(), np.median(tile_cost), np.percentile(tile_cost, 99), tile_cost.max(), tile_cost.max() / tile_cost.mean()))
busy = tile_cost[tile_cost > 4]
print("non-empty tiles %d: mean %.1f max %.1f" % (len(busy), busy.mean(), busy.max()))

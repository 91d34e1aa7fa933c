#!/usr/bin/env python3
"""Diagnostic: beam kernel against the packet walk and the oracle at the bench's size; prints where films differ."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from oracle import binding as ob
from raytrace_amd import _lib, abi, phys, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
level = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _lib.load()
sc = scenes.scan(level=level, dx=3840, dy=2160, spp=spp, tex_size=1024)
o = sc.RenderOptions


def film(dev):
    f = torch.zeros((o.Dy, o.Dx, 3), dtype=torch.float64, device="cuda")
    cd = phys.camera_desc(sc.Camera[0])
    ro = phys.render_opts(o, abi.MODE_FAST, 0, 1, torch.cuda.current_stream().cuda_stream, 0, spp)
    st = abi.Stats()
    _lib.check(lib.rt_render_film(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(f.data_ptr()), C.byref(st)), lib)
    torch.cuda.synchronize()
    return f.cpu().numpy(), st


os.environ["RT_NO_PIXEL_CULL"] = "1"
with phys.DeviceScene(sc) as dev:
    beam, sb = film(dev)
    os.environ["RT_NO_BEAM"] = "1"
    packet, sp = film(dev)
    del os.environ["RT_NO_BEAM"]
d = np.abs(beam - packet)
bad = d.max(axis=2) > 0
print("pixels differing:", int(bad.sum()), "of", bad.size, "max abs", d.max(), "max rel", (d / np.maximum(np.abs(packet), 1e-300)).max())
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:12]:
    print("  px", x, y, "beam", beam[y, x], "packet", packet[y, x])
with ob.OracleScene(sc) as orc:
    img, st, _, ref = orc.render(sc.Camera[0], o, n_threads=16, tile_step=16, want_film=True)
mask = img[..., 3] == 255
for name, f in (("beam", beam), ("packet", packet)):
    err = np.abs(f[mask] - ref[mask]) / np.maximum(np.abs(ref[mask]), 1.0)
    print(name, "vs oracle on", int(mask.sum()), "pixels: max rel", err.max(), "pixels > 1e-12:", int((err.max(axis=1) > 1e-12).sum()))
print("differing pixels inside the oracle's sample:", int((bad & mask).sum()))

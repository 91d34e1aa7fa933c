#!/usr/bin/env python3
"""Times rt_scene_create of the 4K scan scene for 1, 2, 4 and 8 replicas (rt_init with that many ordinals; on a machine with
fewer GPUs the ordinals repeat), and one 1024-spp frame divided among them by rows.  usage: replica_create.py [level]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytrace_amd import _lib, abi, phys, scenes

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
sc = scenes.scan(level=level, dx=3840, dy=2160, spp=1024, tex_size=4096)
gpus = max(1, torch.cuda.device_count())
for n in (1, 2, 4, 8):
    devices = [i % gpus for i in range(n)]
    _lib.init(lib, devices)
    t0 = time.perf_counter()
    dev = phys.DeviceScene(sc)
    t1 = time.perf_counter()
    art = dev.render(sc.Camera[0], sc.RenderOptions, abi.MODE_FAST, abi.SPLIT_ROWS)
    t2 = time.perf_counter()
    art = dev.render(sc.Camera[0], sc.RenderOptions, abi.MODE_FAST, abi.SPLIT_ROWS)
    t3 = time.perf_counter()
    dev.close()
    print(f"{n} replica(s) on devices {devices}: rt_scene_create {t1 - t0:.3f} s, first frame {t2 - t1:.3f} s, second frame {t3 - t2:.3f} s "
          f"({art.Stats.TotalRays / (t3 - t2) / 1e9:.1f} Grays/s)", flush=True)
_lib.init(lib, [0])

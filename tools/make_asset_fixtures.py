#!/usr/bin/env python3
"""Copies the reference's DATA inputs of the benchmark configurations into raytrace_amd/assets/ (run once, in the
container that holds /root/reference; the GPU box only reads the results):

  asset/teapot.obj                         -> teapot.obj.gz     (the OBJ text, gzip: 3644 v / 6320 f, no vt / vn)
  example/material/texture.png, faces.png  -> as they are       (8-bit RGBA PNGs)
  example/charuco/asset/Square150MM.png    -> as it is
  example/gem/led.go (the LEDS table)      -> pal257_leds.json  (numbers only: x, y in mm, ring, PCB and bit index)
"""
import gzip
import json
import os
import re
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytrace_amd", "assets")
os.makedirs(OUT, exist_ok=True)
with open(os.path.join(REF, "asset", "teapot.obj"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "teapot.obj.gz"), "wb", mtime=0) as g:
    g.write(f.read())
for src in ("example/material/texture.png", "example/material/faces.png", "example/charuco/asset/Square150MM.png"):
    shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, os.path.basename(src)))
leds = []
pat = re.compile(r"POSX:\s*(-?[\d.]+),\s*POSY:\s*(-?[\d.]+),\s*PCBN:\s*(\d+),\s*BITN:\s*(\d+),\s*RING:\s*(\d+)")
for line in open(os.path.join(REF, "example", "gem", "led.go")):
    m = pat.search(line)
    if m:
        leds.append({"x_mm": float(m.group(1)), "y_mm": float(m.group(2)), "pcb": int(m.group(3)), "bit": int(m.group(4)), "ring": int(m.group(5))})
assert len(leds) == 257, len(leds)
json.dump({"source": "example/gem/led.go:16-273 (LEDS), positions in mm from the centre of the PAL257 emitter", "leds": leds},
          open(os.path.join(OUT, "pal257_leds.json"), "w"), indent=0)
print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})

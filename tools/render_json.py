#!/usr/bin/env python3
"""Renders a scene in the reference's JSON format (Scene.MarshalJSON, phys/scene.go:82-114; also the body of the
playground's POST /raytrace/render) on the GPU and writes a PNG, as the playground does with its render
(playground/main.go:66-110).  --native hands the text to the library itself (rt_scene_create_from_json: no Python scene is built;
a scene whose textures name image files instead of carrying pixels needs the default route).
usage: render_json.py scene.json out.png [--strict] [--assets DIR] [--reference] [--native]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytrace_amd import abi, obj, phys

ap = argparse.ArgumentParser()
ap.add_argument("scene")
ap.add_argument("out")
ap.add_argument("--strict", action="store_true", help="read like the reference's own UnmarshalJSON (its type registry, its errors)")
ap.add_argument("--assets", default=None, help="directory of the image files a TextureImage without pixels names")
ap.add_argument("--reference", action="store_true", help="float64 reference-order traversal instead of the fast mode")
ap.add_argument("--native", action="store_true", help="let the library read the JSON (rt_scene_create_from_json)")
a = ap.parse_args()
mode = abi.MODE_REFERENCE if a.reference else abi.MODE_FAST
with open(a.scene, "rb") as f:
    text = f.read()
t0 = time.perf_counter()
if a.native:
    dev, cameras, options = phys.DeviceScene.from_json(text)
    if not cameras:
        sys.exit("invalid scene: no cameras in the scene")
    with dev:
        art = dev.render(cameras[0], options, mode)
else:
    scene = phys.Scene.UnmarshalJSON(text, strict=a.strict, asset_dir=a.assets or os.path.dirname(os.path.abspath(a.scene)))
    art = phys.Render(scene, mode)  # validates the scene first, like phys.Render
dt = time.perf_counter() - t0
with open(a.out, "wb") as f:
    f.write(obj.encode_png_rgba8(art.Image))
st = art.Stats
print(f"{st.Dx}x{st.Dy}: {st.TotalRays} rays ({st.RaysLeftScene} left the scene, {st.RaysExceededDepth} exceeded the depth) in {dt:.3f} s -> {a.out}")

#!/usr/bin/env python3
"""Writes tests/golden/oracle_hits.npz and oracle_images.npz from the CPU oracle.

These fixtures are NOT outputs of the reference (it cannot run here): they freeze the oracle's answers
for the parts of the path that no table of the reference pins (BVH traversal, spheres, cameras, materials),
so that a later change to the oracle or to the GPU code that moves them is noticed.  Re-run only on purpose."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import binding as ob  # noqa: E402
from raytrace_amd import phys, scenes  # noqa: E402


def hits_fixture():
    faces = scenes.random_faces(2000, seed=11)
    sc = phys.Scene(Camera=[], Node=[phys.Node("m", phys.NewMesh(faces), phys.DebugNormal()),
                                    phys.Node("s", phys.Sphere((10.0, -20.0, 5.0), 30.0), phys.DebugUV())])
    o, d = scenes.random_rays(512, seed=12)
    with ob.OracleScene(sc) as orc:
        h = orc.intersect(o, d, 1e-3)
    return dict(o=o, d=d, node=h["node"], prim=h["prim"], t=h["t"], at=h["at"], normal=h["normal"], uv=h["uv"],
                boxes=h["boxes_tested"], prims=h["prims_tested"])


def images_fixture():
    out = {}
    for name, sc in (("cornell", scenes.cornell(32, 32, 16, 5)), ("material", scenes.material(48, 48, 4, 5, tex_size=(32, 32))),
                     ("scan", scenes.scan(level=3, dx=48, dy=32, spp=2, tex_size=32)),
                     ("charuco", scenes.charuco(dx=48, dy=36, spp=1))):
        with ob.OracleScene(sc) as orc:
            img, st, _, _ = orc.render(sc.Camera[0], sc.RenderOptions, n_threads=4)
        out[name] = img
        out[name + "_stats"] = np.array([st.total_rays, st.rays_exceeded_depth, st.rays_left_scene, st.shadow_rays], dtype=np.uint64)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_hits.npz"), **hits_fixture())
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **images_fixture())
    print("written")

"""Scene definitions for the benchmark / parity configurations.

These play the role of the reference's example programs (example/*/main.go): they
only build `phys.Scene` values.  Where the reference example is an empty stub
(example/cornell/main.go:4-6, example/teapot/main.go:4-6) or needs an asset that
is not in the tree (the einstein scan, texture PNGs), the scene is defined here
with synthetic, seeded inputs of the same shape; every such substitution is
stated in the function's docstring.  The reference's own data files (teapot.obj, the
PNG textures of example/material and example/charuco, the PAL257 LED table) are used
where they exist: copies live in raytrace_amd/assets.
"""
from __future__ import annotations

import functools
import gzip
import json
import math
import os

import numpy as np

from . import phys
from .phys import MM, M


ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def asset_bytes(name: str) -> bytes:
    """A data file of the reference kept under raytrace_amd/assets (tools/make_asset_fixtures.py says which is which)."""
    with open(os.path.join(ASSETS, name), "rb") as f:
        data = f.read()
    return gzip.decompress(data) if name.endswith(".gz") else data


@functools.lru_cache(maxsize=None)
def asset_image(name: str) -> np.ndarray:
    """A decoded image asset, shared by every scene that uses it (read-only)."""
    from . import obj
    img = obj.decode_png(asset_bytes(name))
    img.setflags(write=False)
    return img


def teapot_faces() -> np.ndarray:
    """The faces of the reference's asset/teapot.obj (3644 vertices, 6320 triangles, no vt / vn, raw units: about
    [-3, 3.43] x [0, 3.15] x [-2, 2]) as ConvertObjectToShape builds them (node.go:195-232): (n, 3, 5) float64, uv = 0."""
    from . import obj
    parsed = obj.ParseFS(obj.MapFS({"teapot.obj": asset_bytes("teapot.obj.gz")}), "teapot.obj")
    return phys.ConvertObjectToShape(parsed).Face


def pal257_led_positions() -> np.ndarray:
    """(257, 2) LED positions in mm from the centre of the PAL257 emitter (example/gem/led.go:16-273)."""
    leds = json.loads(asset_bytes("pal257_leds.json"))["leds"]
    return np.array([(d["x_mm"], d["y_mm"]) for d in leds], dtype=np.float64)


# ---- props (phys/prop.go) -----------------------------------------------------------


def PropSkySphere(radius, mat):  # prop.go:11-19
    return phys.Node("SkySphere", phys.Sphere((0.0, 0.0, 0.0), radius), mat)


def PropAxes(origin, radius, length, prefix=""):  # prop.go:21-75
    if radius <= 0:
        raise ValueError(f"PropAxes: invalid radius {radius} (must be > 0)")
    if length <= 0:
        raise ValueError(f"PropAxes: invalid len {length} (must be > 0)")
    if length <= radius:
        raise ValueError(f"PropAxes: invalid radius {radius} (must be << len {length})")
    out = []
    for name, axis in (("AxisX", (1, 0, 0)), ("AxisY", (0, 1, 0)), ("AxisZ", (0, 0, 1))):
        out.append(phys.Node(prefix + name, phys.Cylinder(origin, axis, radius, length),
                             phys.Emitter(phys.TextureUniform(tuple(float(a) for a in axis)))))
    return out


# ---- deterministic procedural inputs ---------------------------------------------------


def _hash_u32(x: np.ndarray) -> np.ndarray:
    """Integer hash (lowbias32), vectorised; platform independent."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, iz, seed):
    h = _hash_u32(ix.astype(np.uint32) * np.uint32(73856093) ^ iy.astype(np.uint32) * np.uint32(19349663)
                  ^ iz.astype(np.uint32) * np.uint32(83492791) ^ np.uint32(seed))
    return h.astype(np.float64) / 4294967295.0


def value_noise3(p: np.ndarray, seed: int) -> np.ndarray:
    """Trilinear (smoothstep) value noise in [0,1] at points p (n,3)."""
    f = np.floor(p)
    i = f.astype(np.int64)
    t = p - f
    t = t * t * (3 - 2 * t)
    out = np.zeros(len(p))
    for dx in (0, 1):
        wx = t[:, 0] if dx else 1 - t[:, 0]
        for dy in (0, 1):
            wy = t[:, 1] if dy else 1 - t[:, 1]
            for dz in (0, 1):
                wz = t[:, 2] if dz else 1 - t[:, 2]
                out += wx * wy * wz * _lattice(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed)
    return out


def icosphere(level: int):
    """Unit icosphere: 20 * 4**level triangles.  Returns (vertices (n,3), triangles (m,3))."""
    phi = (1 + math.sqrt(5)) / 2
    v = np.array([[-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0], [0, -1, phi], [0, 1, phi], [0, -1, -phi],
                  [0, 1, -phi], [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
                  [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * (len(v) + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    return v, f


def faces_from(vertices: np.ndarray, uv: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """(m,3,5) face array (x, y, z, u, v per vertex) -- the layout of Go's []phys.Face."""
    out = np.empty((len(tris), 3, 5), dtype=np.float64)
    out[:, :, :3] = vertices[tris]
    out[:, :, 3:] = uv[tris]
    return out


def scan_standin_mesh(level: int = 7, seed: int = 1234) -> np.ndarray:
    """Stand-in for the missing einstein photogrammetry scan: an icosphere displaced
    radially by 4 octaves of value noise into a bust-like blob of radius ~85..125
    raw OBJ units, with per-vertex equirectangular UVs.  20*4**level faces
    (level 7: 327,680; level 8: 1,310,720)."""
    v, f = icosphere(level)
    disp = np.zeros(len(v))
    amp, freq = 1.0, 1.5
    for octave in range(4):
        disp += amp * (value_noise3(v * freq + 17.0, seed + octave) - 0.5)
        amp *= 0.5
        freq *= 2.0
    r = 105.0 * (1.0 + 0.35 * disp)
    # a "neck": squeeze below the equator so the silhouette is not a ball
    squeeze = 1.0 - 0.25 * np.clip(-v[:, 2] - 0.3, 0, 1)
    p = v * r[:, None]
    p[:, 0] *= squeeze
    p[:, 1] *= squeeze
    u = (np.arctan2(v[:, 1], v[:, 0]) + math.pi) / (2 * math.pi)
    w = 1.0 - np.arccos(np.clip(v[:, 2], -1, 1)) / math.pi
    return faces_from(p, np.stack([u, w], 1), f)


def procedural_texture(size: int = 4096, seed: int = 5678) -> np.ndarray:
    """(size, size, 4) uint8 NRGBA, opaque: smooth colour fields plus per-texel grain."""
    y, x = np.mgrid[0:size, 0:size]
    fx, fy = x / size, y / size
    img = np.empty((size, size, 4), dtype=np.uint8)
    grain = (_hash_u32((x.astype(np.uint32) * np.uint32(size) + y.astype(np.uint32)) ^ np.uint32(seed)) & 31).astype(np.float64)
    for c, (ax, ay, ph) in enumerate(((6.0, 4.0, 0.3), (3.0, 7.0, 1.1), (5.0, 5.0, 2.0))):
        base = 0.5 + 0.35 * np.sin(2 * math.pi * (ax * fx + ph)) * np.cos(2 * math.pi * (ay * fy + 0.5 * ph))
        img[..., c] = np.clip(base * 224.0 + grain, 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


# ---- configurations -----------------------------------------------------------------------


def cornell(dx=256, dy=256, spp=64, depth=5, seed=0) -> phys.Scene:
    """Cornell box (the reference example is an empty stub): 5 Lambertian quads of a
    555 mm cube, an Emitter quad (130 mm, colour 4,4,4) under the ceiling, two
    Lambertian spheres, one point light so next-event estimation runs; pinhole camera."""
    s = 555.0 * MM
    h = s / 2
    white = phys.Lambertian(phys.TextureUniform((0.73, 0.73, 0.73)))
    red = phys.Lambertian(phys.TextureUniform((0.65, 0.05, 0.05)))
    green = phys.Lambertian(phys.TextureUniform((0.12, 0.45, 0.15)))
    light = phys.Emitter(phys.TextureUniform((4.0, 4.0, 4.0)))
    nodes = [
        phys.Node("floor", phys.Quad((0, -h, 0), (0.0, 1.0, 0.0), s, s), white),
        phys.Node("ceiling", phys.Quad((0, h, 0), (0.0, -1.0, 0.0), s, s), white),
        phys.Node("back", phys.Quad((0, 0, -h), (0.0, 0.0, 1.0), s, s), white),
        phys.Node("left", phys.Quad((-h, 0, 0), (1.0, 0.0, 0.0), s, s), red),
        phys.Node("right", phys.Quad((h, 0, 0), (-1.0, 0.0, 0.0), s, s), green),
        phys.Node("lamp", phys.Quad((0, h - 1.0 * MM, 0), (0.0, -1.0, 0.0), 130.0 * MM, 130.0 * MM), light),
        phys.Node("ball1", phys.Sphere((-110.0 * MM, -h + 90.0 * MM, -60.0 * MM), 90.0 * MM), white),
        phys.Node("ball2", phys.Sphere((120.0 * MM, -h + 70.0 * MM, 80.0 * MM), 70.0 * MM),
                  phys.Lambertian(phys.TextureUniform((0.3, 0.4, 0.8)))),
    ]
    # pinhole: eye on +z looking down -z at the open face
    eye = (0.0, 0.0, 1050.0 * MM)
    plane_z = h
    half = h * (1050.0 * MM - plane_z) / (1050.0 * MM - plane_z)  # image plane = the box opening
    cam = phys.PinholeCamera(LowerLeftCorner=(-half, -half, plane_z), Origin=eye, Horizontal=(2 * half, 0.0, 0.0),
                             Vertical=(0.0, 2 * half, 0.0))
    return phys.Scene(Camera=[cam], Node=nodes,
                      Light=[phys.PointLight((0.0, h - 60.0 * MM, 100.0 * MM), (0.3, 0.3, 0.3))],
                      RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def teapot(dx=1920, dy=1080, spp=256, depth=6, seed=0, aperture=0.0) -> phys.Scene:
    """Teapot config (the reference example is an empty stub): the reference's asset/teapot.obj
    (6320 faces) scaled x10 mm, Metal{0.9, Fuzz 0.05}, on a 400 mm Lambertian checkerboard quad,
    two point lights, FocusableCamera."""
    faces = teapot_faces().copy()
    faces[:, :, :3] *= 10.0 * MM
    mesh = phys.NewMesh(faces)
    checker = phys.TextureCheckerboard(Odd=phys.TextureUniform((0.2, 0.5, 0.6)), Even=phys.TextureUniform((0.9, 0.9, 0.9)),
                                       Frequency=20.0)
    nodes = [
        phys.Node("teapot", mesh, phys.Metal((0.9, 0.9, 0.9), 0.05)),
        phys.Node("ground", phys.Quad((0.0, -0.01 * MM, 0.0), (0.0, 1.0, 0.0), 400.0 * MM, 400.0 * MM),
                  phys.Lambertian(checker)),
    ]
    look_from = (90.0 * MM, 70.0 * MM, 110.0 * MM)
    look_at = (0.0, 12.0 * MM, 0.0)
    wd = math.dist(look_from, look_at)
    cam = phys.FocusableCamera(LookFrom=look_from, LookAt=look_at, VUp=(0.0, 1.0, 0.0),
                               FOVHeight=0.45 * dy / dy, FOVWidth=0.45 * dx / dy, Aperture=aperture, WorkingDistance=wd)
    lights = [phys.PointLight((200.0 * MM, 300.0 * MM, 200.0 * MM), (0.5, 0.5, 0.5)),
              phys.PointLight((-250.0 * MM, 200.0 * MM, 50.0 * MM), (0.2, 0.2, 0.25))]
    return phys.Scene(Camera=[cam], Node=nodes, Light=lights, RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def scan(level=8, dx=3840, dy=2160, spp=1024, depth=10, seed=0, tex_size=4096, lit=False) -> phys.Scene:
    """example/scan with a stand-in for the missing einstein.obj: one Mesh node whose
    material is what ConvertObjectToMaterial builds (node.go:269-309): Emitter with a
    bilinear/repeat image texture, no lights; orthographic camera of
    example/scan/main.go:113-125 at theta = 0 (FOV 250 units high, width scaled by the
    aspect ratio).  lit=True is the "scan-lit" variant: Lambertian + one point light."""
    faces = scan_standin_mesh(level)
    mesh = phys.NewMesh(faces)
    tex = phys.TextureImage(procedural_texture(tex_size), "bilinear", "repeat")
    mat = phys.Lambertian(tex) if lit else phys.Emitter(tex)
    cam = phys.OrthographicCamera(LookFrom=(100.0 * M, 0.0, 100.0 * M), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0),
                                  FOVHeight=250.0, FOVWidth=250.0 * dx / dy)
    lights = [phys.PointLight((5000.0, 5000.0, 5000.0), (0.8, 0.8, 0.8))] if lit else []
    return phys.Scene(Camera=[cam], Node=[phys.Node("default", mesh, mat)], Light=lights,
                      RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def scan_from_obj(fsys, model_path, dx=512, dy=512, spp=1, depth=10, seed=0, step=0, nstep=4) -> phys.Scene:
    """example/scan/main.go:71-150: parse the OBJ (with its MTL and PNG texture) from `fsys`, one Mesh node per
    material (Emitter + texture), no lights, and the orthographic camera number `step` of the example's orbit
    (LookFrom = (100 m cos t, 100 m sin t, 100 m - 80 m sin t), FOV 250 units; the example renders camera 0)."""
    from . import obj
    parsed = obj.ParseFS(fsys, model_path)
    nodes = phys.ConvertObjectToNodes(parsed, fsys)
    cams = []
    for i in range(nstep):
        t = i * 2.0 * math.pi / nstep
        cams.append(phys.OrthographicCamera(LookFrom=(100.0 * M * math.cos(t), 100.0 * M * math.sin(t), 100.0 * M - 80.0 * math.sin(t) * M),
                                            LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0), FOVHeight=250.0, FOVWidth=250.0))
    cams = cams[step % nstep:] + cams[:step % nstep]
    return phys.Scene(Camera=cams, Node=nodes, Light=[], RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def write_obj(faces: np.ndarray, mtl_name="scan.mtl", material="scan", texture="scan.png") -> tuple:
    """(obj text, mtl text) for an (n, 3, 5) face array: unshared v / vt per corner, '%.17g' so that parsing
    gives back the same float64 values."""
    f = np.asarray(faces, dtype=np.float64).reshape(-1, 3, 5)
    lines = [f"mtllib {mtl_name}"]
    for v in f.reshape(-1, 5):
        lines.append("v %.17g %.17g %.17g" % (v[0], v[1], v[2]))
    for v in f.reshape(-1, 5):
        lines.append("vt %.17g %.17g" % (v[3], v[4]))
    lines.append(f"usemtl {material}")
    for i in range(len(f)):
        a = 3 * i + 1
        lines.append(f"f {a}/{a} {a + 1}/{a + 1} {a + 2}/{a + 2}")
    return "\n".join(lines) + "\n", f"newmtl {material}\nKd 1 1 1\nmap_Kd {texture}\n"


def material(dx=2048, dy=2048, spp=4096, depth=5, seed=0, tex_size=None) -> phys.Scene:
    """example/material/main.go:29-123: the sphere grid of Lambertian / Dielectric / Metal /
    debug materials, three axis cylinders, orthographic camera with VUp = X, one point
    light, with the example's own texture.png (568x564) and faces.png (730x726).  tex_size=None
    (default) uses them; a (rows, columns) pair swaps in procedural images of that size."""

    def node(col, row, name, mat):  # main.go:15-26
        cx = col * 10.0 * MM - 45.0 * MM
        cy = row * 10.0 * MM - 25.0 * MM
        return phys.Node(name, phys.Sphere((cx, cy, 0.0), 4.0 * MM), mat)

    def lam(c):
        return phys.Lambertian(phys.TextureUniform(c))

    if tex_size is None:
        tex_a, tex_b = asset_image("texture.png"), asset_image("faces.png")
    else:
        tex_a = procedural_texture(max(tex_size), 11)[:tex_size[0], :tex_size[1]].copy()
        tex_b = procedural_texture(730, 12)[:726, :730].copy()
    nodes = [
        node(0, 0, "checker red white", phys.Lambertian(phys.TextureCheckerboard(
            phys.TextureUniform((0.2, 0.5, 0.6)), phys.TextureUniform((0.5, 0.0, 0.0)), 5.0))),
        node(1, 0, "checker complementary colors", phys.Lambertian(phys.TextureCheckerboard(
            phys.TextureUniform((0.0, 0.0, 123 / 255)), phys.TextureUniform((242 / 255, 0.0, 0.0)), 5.0))),
        node(2, 0, "image texture", phys.Lambertian(phys.TextureImage(tex_a, "bilinear", "repeat"))),
        node(3, 0, "face texture", phys.Lambertian(phys.TextureImage(tex_b, "bilinear", "repeat"))),
    ]
    for col, lvl, v in ((0, "min", 0.05), (1, "med", 0.5), (2, "max", 0.9)):
        nodes.append(node(col, 1, f"Lambertian {lvl} gray", lam((v, v, v))))
        nodes.append(node(col, 2, f"Lambertian {lvl} red", lam((v, 0.0, 0.0))))
        nodes.append(node(col, 3, f"Lambertian {lvl} green", lam((0.0, v, 0.0))))
        nodes.append(node(col, 4, f"Lambertian {lvl} blue", lam((0.0, 0.0, v))))
    for col, lbl, rough in ((3, "clear", 0.0), (4, "good", 0.05), (5, "frosty", 0.2)):
        for row, n in ((1, 1.2), (2, 1.5), (3, 1.8), (4, 2.0)):
            nodes.append(node(col, row, f"dielectric {lbl} n={n}", phys.Dielectric(n, 1.0, rough)))
    nodes += [
        node(6, 1, "shiny metal fine gray", phys.Metal((0.9, 0.9, 0.9), 0.025)),
        node(6, 2, "shiny metal smooth gray", phys.Metal((0.5, 0.5, 0.5), 0.05)),
        node(6, 3, "shiny metal medium gray", phys.Metal((0.5, 0.5, 0.5), 0.1)),
        node(6, 4, "shiny metal rough gray", phys.Metal((0.5, 0.5, 0.5), 0.15)),
        node(7, 1, "metal fine gray", phys.Metal((0.3, 0.3, 0.3), 0.025)),
        node(7, 2, "metal smooth red", phys.Metal((0.3, 0.0, 0.0), 0.05)),
        node(7, 3, "metal medium green", phys.Metal((0.0, 0.3, 0.0), 0.1)),
        node(7, 4, "metal rough blue", phys.Metal((0.0, 0.0, 0.3), 0.15)),
        node(9, 1, "ShaderNormal", phys.DebugNormal()),
        node(9, 2, "ShaderUV", phys.DebugUV()),
    ]
    nodes += PropAxes((-45.0 * MM, -45.0 * MM, 0.0), MM * 0.15, MM * 4)
    cam = phys.OrthographicCamera(LookFrom=(40.0 * MM, 50.0 * MM, 400.0 * MM), LookAt=(0.0, 0.0, 0.0), VUp=(1.0, 0.0, 0.0),
                                  FOVHeight=110.0 * MM, FOVWidth=110.0 * MM)
    return phys.Scene(Camera=[cam], Node=nodes,
                      Light=[phys.PointLight((1000.0 * MM, 1000.0 * MM, 1000.0 * MM), (0.3, 0.3, 0.3))],
                      RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def charuco(dx=1440, dy=1080, spp=16, depth=6, seed=0, frame=0, frames_per_orbit=240, leds=False) -> phys.Scene:
    """example/charuco/main.go:36-289: emitter spheres on the corners of a 150 mm box, three axis cylinders,
    twelve cylinder rods along the box edges, a 150 mm quad with an image texture (default nearest
    sampling), CalibratedCamera{IntrinsicsFireflyDLGeneric6mm} on an orbit of radius 400 mm at height 320 mm
    with VUp = -Z, one point light.  The board is the example's own asset/Square150MM.png (1772x1772).
    frame selects the orbit position (theta = 2 pi frame / frames_per_orbit).
    leds=True adds the light-loop stress case: a Lambertian ground quad under 257 point lights
    (I = 0.3 / 257 each) placed 100 mm above it at the LED positions of example/gem/led.go (PAL257)."""
    W = H = D = 150.0 * MM
    colormap0 = (0.5, 0.5, 0.5)
    P = (203.0 / 255.0, 136.0 / 255.0, 206.0 / 255.0)
    cx_, cy_, cz_ = (1.0, 0.0, 157.0 / 255.0), (157.0 / 255.0, 1.0, 0.0), (0.0, 57.0 / 255.0, 1.0)

    def ball(name, col, x, y, z):  # main.go:342-350
        return phys.Node(name, phys.Sphere((x, y, z), 4.0 * MM), phys.Emitter(phys.TextureUniform(col)))

    def rod(name, col, o, d, h):
        return phys.Node(name, phys.Cylinder(o, d, 1.0 * MM, h), phys.Emitter(phys.TextureUniform(col)))

    nodes = [ball("Origin", colormap0, 0.0, 0.0, 0.0)]
    corners = [(-W / 2, -H / 2), (W / 2, -H / 2), (W / 2, H / 2), (-W / 2, H / 2)]
    for i, (x, y) in enumerate(corners):
        nodes.append(ball(f"P{i + 1}", P, x, y, D))
    for i, (x, y) in enumerate(corners):
        nodes.append(ball(f"P{i + 5}", P, x, y, 0.0))
    nodes += PropAxes((0.0, 0.0, 0.0), 0.5 * MM, 20.0 * MM, "")
    X, Y, Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    nodes += [
        rod("AxisX1Top", cx_, (-W / 2, -H / 2, D), X, W), rod("AxisX2Top", cx_, (-W / 2, H / 2, D), X, W),
        rod("AxisY1Top", cy_, (-W / 2, -H / 2, D), Y, H), rod("AxisY2Top", cy_, (W / 2, -H / 2, D), Y, H),
        rod("AxisX1Bottom", cx_, (-W / 2, -H / 2, 0.0), X, W), rod("AxisX2Bottom", cx_, (-W / 2, H / 2, 0.0), X, W),
        rod("AxisZ1Top", cz_, (-W / 2, H / 2, 0.0), Z, D), rod("AxisZ2Top", cz_, (W / 2, H / 2, 0.0), Z, D),
        rod("AxisZ1Bottom", cz_, (-W / 2, -H / 2, 0.0), Z, D), rod("AxisZ2Bottom", cz_, (W / 2, -H / 2, 0.0), Z, D),
        rod("AxisY1Bottom", cy_, (-W / 2, -H / 2, 0.0), Y, H), rod("AxisY2Bottom", cy_, (W / 2, -H / 2, 0.0), Y, H),
        phys.Node("CharucoSquare150MM", phys.Quad((0.0, 0.0, 0.0), Z, 150.0 * MM, 150.0 * MM),
                  phys.Emitter(phys.TextureImage(asset_image("Square150MM.png"), "", ""))),
    ]
    lights = [phys.PointLight((500.0 * MM, 500.0 * MM, 500.0 * MM), (0.3, 0.3, 0.3))]
    if leds:
        nodes.append(phys.Node("LedGround", phys.Quad((0.0, 0.0, -2.0 * MM), Z, 400.0 * MM, 400.0 * MM),
                               phys.Lambertian(phys.TextureUniform((0.8, 0.8, 0.8)))))
        each = 0.3 / 257.0
        lights = [phys.PointLight((x * MM, y * MM, 100.0 * MM), (each, each, each)) for x, y in pal257_led_positions()]
    theta = 2.0 * math.pi * (frame % frames_per_orbit) / frames_per_orbit
    cam = phys.CalibratedCamera(
        Intrinsics=phys.IntrinsicsFireflyDLGeneric6mm,
        Extrinsics=phys.CameraExtrinsics(LookFrom=(400.0 * MM * math.cos(theta), 400.0 * MM * math.sin(theta), 320.0 * MM),
                                         LookAt=(0.0, 0.0, D / 2), VUp=(0.0, 0.0, -1.0)))
    return phys.Scene(Camera=[cam], Node=nodes, Light=lights, RenderOptions=phys.RenderOptions(seed, spp, depth, dx, dy))


def random_faces(n: int, seed: int = 1) -> np.ndarray:
    """The reference's BVH micro-benchmark input (shape_bvh_test.go:60-98): n faces with
    vertices uniform in [-50,50]^3 and uniform UVs (seeded here; the reference uses the
    unseeded global generator)."""
    rng = np.random.default_rng(seed)
    f = rng.random((n, 3, 5))
    f[:, :, :3] = f[:, :, :3] * 100.0 - 50.0
    return f


def random_rays(n: int, seed: int = 2):
    """shape_bvh_test.go:101-130: origins uniform in [-100,100]^3, uniform directions."""
    rng = np.random.default_rng(seed)
    o = rng.random((n, 3)) * 200.0 - 100.0
    theta = rng.random(n) * 2 * math.pi
    ph = np.arccos(2 * rng.random(n) - 1)
    d = np.stack([np.sin(ph) * np.cos(theta), np.sin(ph) * np.sin(theta), np.cos(ph)], 1)
    return o, d

"""GPU parity tests: the HIP library (through its C ABI) against the CPU oracle.

Bars
  * reference mode (float64, reference visit order): node / face ids, t, hit point, normal,
    uv and the per-ray counts of box and face tests are compared bit for bit;
  * fast mode: face ids may differ only where two faces tie; t must still be exact because
    faces are intersected in float64 with the reference's arithmetic;
  * images: identical random streams on both sides, so low-spp images are compared pixel by
    pixel (<= 1 level apart except for a small fraction where a last-bit libm difference in
    sin / cos / pow flips a path) and counters are compared to 0.1 %.
"""
import json
import os

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
MAXF = 1.7976931348623157e308


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load()
    return binding


def mesh_scene(faces):
    return phys.Scene(Camera=[], Node=[phys.Node("m", phys.NewMesh(faces), phys.DebugNormal())])


def assert_hits_identical(a, b, what):
    assert np.array_equal(a["node"], b["node"]), what
    hit = a["node"] >= 0
    for f in ("prim", "t", "at", "normal", "uv"):
        assert np.array_equal(a[f][hit], b[f][hit]), (what, f)


def test_library_reports_gfx950_and_loads():
    from raytrace_amd import _lib
    lib = _lib.load()
    assert lib is not None


@pytest.mark.parametrize("n_faces,n_rays", [(1, 2000), (4, 2000), (5, 2000), (1000, 20000), (100000, 100000)])
def test_reference_mode_random_faces_bit_exact(ob, n_faces, n_rays):
    # the reference's own BVH micro-benchmark input shape (shape_bvh_test.go:60-130), seeded
    sc = mesh_scene(scenes.random_faces(n_faces, seed=n_faces))
    o, d = scenes.random_rays(n_rays, seed=7)
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        g = dev.intersect(o, d, 0.001, MAXF, abi.MODE_REFERENCE)
        c = orc.intersect(o, d, 0.001, MAXF)
        assert_hits_identical(g, c, f"random faces {n_faces}")
        assert np.array_equal(g["boxes_tested"], c["boxes_tested"])
        assert np.array_equal(g["prims_tested"], c["prims_tested"])
        gi, ci = dev.bvh_info(0), orc.bvh_info(0)
        assert (gi.n_nodes, gi.n_leaves, gi.max_depth, gi.max_leaf_size) == (ci.n_nodes, ci.n_leaves, ci.max_depth, ci.max_leaf_size)
        if n_faces >= 1000:
            assert (g["node"] >= 0).sum() > 100


def scan_rays(sc, n, seed):
    """Primary rays of the scan camera plus incoherent rays aimed into the mesh."""
    rng = np.random.default_rng(seed)
    st = rng.random((n, 2))
    from oracle import binding
    o, d = binding.cast(sc.Camera[0], st, np.zeros((n, 2), dtype=np.uint32))
    o2 = rng.normal(size=(n, 3))
    o2 = o2 / np.linalg.norm(o2, axis=1, keepdims=True) * 400.0
    tgt = rng.normal(size=(n, 3)) * 40.0
    d2 = tgt - o2
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    return np.concatenate([o, o2]), np.concatenate([d, d2])


def test_reference_mode_scan_mesh_bit_exact(ob):
    sc = scenes.scan(level=5, dx=64, dy=64, spp=1, tex_size=64)
    o, d = scan_rays(sc, 20000, 3)
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        g = dev.intersect(o, d, 1e-6, MAXF, abi.MODE_REFERENCE)
        c = orc.intersect(o, d, 1e-6, MAXF)
        assert_hits_identical(g, c, "scan level 5")
        assert np.array_equal(g["boxes_tested"], c["boxes_tested"])
        assert np.array_equal(g["prims_tested"], c["prims_tested"])
        assert (g["node"] >= 0).mean() > 0.3


@pytest.mark.parametrize("case", ["scan", "random"])
def test_fast_mode_hits_match_reference_t_exact(ob, case):
    if case == "scan":
        sc = scenes.scan(level=6, dx=64, dy=64, spp=1, tex_size=64)
        o, d = scan_rays(sc, 50000, 5)
        tmin = 1e-6
    else:
        sc = mesh_scene(scenes.random_faces(200000, seed=9))
        o, d = scenes.random_rays(100000, seed=11)
        tmin = 0.001
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        g = dev.intersect(o, d, tmin, MAXF, abi.MODE_FAST)
        c = orc.intersect(o, d, tmin, MAXF)
    assert np.array_equal(g["node"], c["node"]), "hit / miss must agree exactly"
    hit = c["node"] >= 0
    assert np.array_equal(g["t"][hit], c["t"][hit]), "t is computed in float64 with the reference's arithmetic"
    same = g["prim"][hit] == c["prim"][hit]
    # a different face id is only legitimate when two faces give the same t (shared edge / vertex)
    assert same.mean() > 0.9999, same.mean()
    assert np.array_equal(g["normal"][hit][same], c["normal"][hit][same])
    assert np.array_equal(g["uv"][hit][same], c["uv"][hit][same])


def test_fast_mode_unrelated_rays_in_a_scene_of_meshes_among_other_shapes(ob):
    """A batch of unrelated rays in fast mode: every lane lists the nodes its own ray reaches, and the nodes that hold a mesh
    (plain or under a transform) are still entered by all their lanes together.  Three meshes between spheres, quads and a
    cylinder; the oracle decides node, distance and attributes."""
    rng = np.random.default_rng(21)
    v, t = scenes.icosphere(3)
    uv = np.abs(v[:, :2])

    def ball(center, radius):
        return phys.NewMesh(scenes.faces_from(v * radius + np.asarray(center), uv, t))

    turn = phys.Transform((0.0, -2.5, 0.3), phys.RotationMatrixZ(0.9), (1.0, 0.6, 1.2))
    mat = phys.DebugNormal()
    nodes = [phys.Node("q0", phys.Quad((0.0, 0.0, -1.5), (0.0, 0.0, 1.0), 14.0, 14.0), mat),
             phys.Node("m0", ball((2.5, 0.0, 0.0), 0.9), mat),
             phys.Node("s0", phys.Sphere((0.0, 0.0, 0.0), 0.7), mat),
             phys.Node("s1", phys.Sphere((-2.5, 2.0, 0.5), 0.6), mat),
             phys.Node("m1", phys.TransformedShape(ball((0.0, 0.0, 0.0), 0.8), turn), mat),
             phys.Node("c0", phys.Cylinder((-2.5, -2.0, -1.0), (0.0, 0.0, 1.0), 0.5, 1.5), mat),
             phys.Node("m2", ball((0.0, 2.8, 0.4), 0.7), mat),
             phys.Node("q1", phys.Quad((3.5, 3.5, 0.0), (-1.0, -1.0, 0.3), 2.0, 2.0), mat)]
    sc = phys.Scene(Camera=[], Node=nodes)
    n = 60000
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(4.0, 9.0, size=(n, 1))
    d = rng.uniform(-3.5, 3.5, size=(n, 3)) * np.array([1.0, 1.0, 0.4]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        g = dev.intersect(o, d, 1e-6, MAXF, abi.MODE_FAST)
        c = orc.intersect(o, d, 1e-6, MAXF)
    assert np.array_equal(g["node"], c["node"])
    for k in range(len(nodes)):
        assert (c["node"] == k).sum() > 50, nodes[k].Name
    hit = c["node"] >= 0
    assert np.array_equal(g["t"][hit], c["t"][hit])
    same = g["prim"][hit] == c["prim"][hit]
    assert same.mean() > 0.999, same.mean()
    assert np.array_equal(g["normal"][hit][same], c["normal"][hit][same])
    round_shape = np.isin(c["node"][hit][same], [2, 3, 5])  # sphere / cylinder coordinates go through atan2 and acos: the last bit is libm's
    assert np.array_equal(g["uv"][hit][same][~round_shape], c["uv"][hit][same][~round_shape])
    assert np.allclose(g["uv"][hit][same][round_shape], c["uv"][hit][same][round_shape], rtol=0, atol=1e-12)


def test_fast_mode_drops_never_hit_flat_subtrees(ob):
    # AABB.hit rejects tmax <= tmin (shape.go:91): a leaf whose faces are coplanar with an axis
    # plane has a zero-thickness box and is never entered by the reference.  Fast mode must agree.
    flat = np.zeros((2, 3, 5))
    flat[0, :, :3] = [[-1, 0, -1], [1, 0, -1], [1, 0, 1]]
    flat[1, :, :3] = [[-1, 0, -1], [1, 0, 1], [-1, 0, 1]]
    sc = mesh_scene(flat)
    o = np.array([[0.2, 1.0, 0.1], [0.5, 2.0, -0.3]])
    d = np.array([[0.0, -1.0, 0.0], [0.1, -1.0, 0.05]])
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        c = orc.intersect(o, d, 1e-6, MAXF)
        assert (c["node"] == -1).all()
        for mode in (abi.MODE_REFERENCE, abi.MODE_FAST):
            assert (dev.intersect(o, d, 1e-6, MAXF, mode)["node"] == -1).all()


def test_analytic_shapes_golden_tables_on_gpu(ob):
    def load(name):
        with open(os.path.join(G, name)) as f:
            return json.load(f)

    def check(shape, o, d, tmin, tmax, want_hit, point=None, normal=None, uv=None, name=""):
        sc = phys.Scene(Camera=[], Node=[phys.Node("s", shape, phys.DebugNormal())])
        with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
            for mode in (abi.MODE_REFERENCE, abi.MODE_FAST):
                h = dev.intersect([o], [d], tmin, tmax, mode)
                c = orc.intersect([o], [d], tmin, tmax)
                assert (h["node"][0] >= 0) == want_hit, name
                assert_hits_identical(h, c, name)
                if want_hit:
                    assert np.allclose(h["at"][0], point, atol=1e-6, rtol=0), name
                    assert np.allclose(h["normal"][0], normal, atol=1e-6, rtol=0), name
                    if uv is not None:
                        assert np.allclose(h["uv"][0], uv, atol=1e-6, rtol=0), name

    g = load("triangle_collide.json")
    for grp in g["groups"]:
        for c in grp["cases"]:
            want = c.get("code_hit", c["expect_hit"])
            check(phys.Triangle(*grp["triangle"]), c["o"], c["d"], 0.0, MAXF, want, c.get("point"), c.get("normal"), name=c["name"])
    g = load("cylinder.json")
    cyl = phys.Cylinder(g["cylinder"]["origin"], g["cylinder"]["direction"], g["cylinder"]["radius"], g["cylinder"]["height"])
    for c in g["cases"]:
        dd = np.array(c["d"], dtype=np.float64)
        check(cyl, c["o"], dd / np.linalg.norm(dd), 0.0001, MAXF, c["hit"], c.get("point"), c.get("normal"), name=c["name"])
    g = load("quad.json")
    q = phys.Quad(g["quad"]["center"], g["quad"]["normal"], g["quad"]["width"], g["quad"]["height"])
    for c in g["cases"]:
        check(q, c["o"], c["d"], 1e-6, 1000.0, c["hit"], c.get("point"), c.get("normal"), c.get("code_uv", c.get("expected_uv")), name=c["name"])


def test_mixed_scene_random_rays_bit_exact(ob):
    # every shape kind, including a transformed sphere, a transformed mesh and a nested transform
    rng = np.random.default_rng(21)
    tf = phys.Transform((3.0, -1.0, 2.0), phys.RotationMatrixY(0.7), (1.5, 0.5, 2.0))
    tf2 = phys.Transform((-4.0, 2.0, 0.0), phys.RotationMatrixZ(-0.4), (0.8, 0.8, 0.8))
    mesh = phys.NewMesh(scenes.random_faces(300, seed=4) * np.array([0.05, 0.05, 0.05, 1, 1]))
    nodes = [
        phys.Node("s0", phys.Sphere((0, 0, 0), 1.0), phys.DebugUV()),
        phys.Node("s1", phys.Sphere((2.5, 0.5, -1), 0.7, abi.UVMAP_BOX), phys.DebugUV()),
        phys.Node("s2", phys.Sphere((-2.5, 0.5, 1), 0.7, abi.UVMAP_EQUIRECT_SOUTH_UP), phys.DebugUV()),
        phys.Node("q", phys.Quad((0, -2, 0), (0.0, 1.0, 0.0), 8.0, 6.0), phys.DebugUV()),
        phys.Node("q2", phys.Quad((0, 0, -4), (1.0, 0.0, 0.0), 3.0, 2.0), phys.DebugUV()),
        phys.Node("c", phys.Cylinder((-1, -1, 2), (0.0, 0.6, 0.8), 0.5, 2.0), phys.DebugUV()),
        phys.Node("t", phys.Triangle((4, 0, 0), (5, 0, 1), (4, 1.5, 0.5)), phys.DebugUV()),
        phys.Node("tuv", phys.TriangleUV((-4, 0, 0), (-5, 0, 1), (-4, 1.5, 0.5), (0, 0), (1, 0), (0, 1),
                                         tuple(np.cross([-1, 0, 1], [0, 1.5, 0.5]) / np.linalg.norm(np.cross([-1, 0, 1], [0, 1.5, 0.5])))),
                  phys.DebugUV()),
        phys.Node("f", phys.Face([[0, 3, 0, 0.1, 0.2], [1, 3, 0, 0.9, 0.1], [0, 4, 1, 0.3, 0.8]]), phys.DebugUV()),
        phys.Node("ts", phys.TransformedShape(phys.Sphere((0, 0, 0), 1.0), tf), phys.DebugUV()),
        phys.Node("tm", phys.TransformedShape(mesh, tf2), phys.DebugUV()),
        phys.Node("tt", phys.TransformedShape(phys.TransformedShape(phys.Cylinder((0, 0, 0), (0.0, 1.0, 0.0), 0.4, 1.0), tf), tf2), phys.DebugUV()),
        phys.Node("m", mesh, phys.DebugUV()),
    ]
    sc = phys.Scene(Camera=[], Node=nodes)
    n = 40000
    o = rng.normal(size=(n, 3)) * 6.0
    tgt = rng.normal(size=(n, 3)) * 2.5
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        c = orc.intersect(o, d, 1e-6, MAXF)
        g = dev.intersect(o, d, 1e-6, MAXF, abi.MODE_REFERENCE)
        hit = c["node"] >= 0
        assert set(np.unique(c["node"][hit])) == set(range(len(nodes))), "every node should be hit by some ray"
        assert np.array_equal(g["node"], c["node"])
        assert np.array_equal(g["prim"][hit], c["prim"][hit])
        assert np.array_equal(g["t"][hit], c["t"][hit])
        assert np.array_equal(g["at"][hit], c["at"][hit])
        assert np.array_equal(g["normal"][hit], c["normal"][hit])
        # uv of spheres goes through atan2 / acos, whose last bit differs between glibc and the GPU's libm
        assert np.allclose(g["uv"][hit], c["uv"][hit], atol=1e-12, rtol=0)
        f = dev.intersect(o, d, 1e-6, MAXF, abi.MODE_FAST)
        assert np.array_equal(f["node"], c["node"])
        assert np.array_equal(f["t"][hit], c["t"][hit])


def test_cast_batch_all_cameras(ob):
    import ctypes as C
    from raytrace_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    n = 5000
    st = rng.random((n, 2))
    ps = np.stack([rng.integers(0, 1 << 20, n), rng.integers(0, 4096, n)], 1).astype(np.uint32)
    cams = [
        phys.PinholeCamera((-2.0, -1.0, -1.0), (0.1, 0.2, 0.3), (4.0, 0.0, 0.0), (0.0, 2.0, 0.0)),
        phys.OrthographicCamera((1e11, 0.0, 1e11), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 250.0, 444.0),
        phys.FocusableCamera((90e6, 70e6, 110e6), (0.0, 12e6, 0.0), (0.0, 1.0, 0.0), 0.45, 0.8, 2e6, 1.5e8),
        phys.FocusableCamera((90e6, 70e6, 110e6), (0.0, 12e6, 0.0), (0.0, 1.0, 0.0), 0.45, 0.8, 0.0, 1.5e8),
        phys.CalibratedCamera(phys.IntrinsicsFireflyDLGeneric6mm, phys.CameraExtrinsics((4e8, 0.0, 3.2e8), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))),
        phys.CalibratedCamera(phys.IntrinsicsFireflyDLComputar16mm, phys.CameraExtrinsics((4e8, 1e8, 3.2e8), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))),
    ]
    for cam in cams:
        co, cd = ob.cast(cam, st, ps, seed=5)
        go, gd = np.zeros((n, 3)), np.zeros((n, 3))
        desc = phys.camera_desc(cam)
        dp = C.POINTER(C.c_double)
        _lib.check(lib.rt_cast_batch(C.byref(desc), st.ctypes.data_as(dp), ps.ctypes.data_as(C.POINTER(C.c_uint32)), n, 5,
                                     go.ctypes.data_as(dp), gd.ctypes.data_as(dp)), lib)
        assert np.array_equal(go, co), type(cam).__name__
        assert np.array_equal(gd, cd), type(cam).__name__


def _render_pair(ob, sc, mode):
    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:
        art = dev.render(sc.Camera[0], sc.RenderOptions, mode)
        img, st, _, _ = orc.render(sc.Camera[0], sc.RenderOptions)
    return art, img, st


@pytest.mark.parametrize("mode", [abi.MODE_REFERENCE, abi.MODE_FAST])
def test_render_scan_matches_oracle(ob, mode):
    # deterministic path: one primary ray per pixel, Emitter + bilinear texture, no bounce
    sc = scenes.scan(level=5, dx=320, dy=180, spp=1, tex_size=256)
    art, img, st = _render_pair(ob, sc, mode)
    assert art.Stats.TotalRays == st.total_rays == 320 * 180
    assert art.Stats.RaysLeftScene == st.rays_left_scene
    assert art.Stats.RaysExceededDepth == st.rays_exceeded_depth == 0
    assert np.array_equal(art.Image[..., 3], img[..., 3])
    diff = np.abs(art.Image.astype(int) - img.astype(int))
    assert diff.max() <= 1
    assert (diff > 0).mean() < 1e-4
    assert img[..., :3].max() > 100


@pytest.mark.parametrize("name", ["cornell", "material", "teapot", "scan_lit"])
def test_render_stochastic_scenes_match_oracle_sample_for_sample(ob, name):
    sc = {
        "cornell": lambda: scenes.cornell(96, 96, 8, 5),
        "material": lambda: scenes.material(160, 160, 4, 5, tex_size=(64, 64)),
        "teapot": lambda: scenes.teapot(160, 90, 4, 6),
        "scan_lit": lambda: scenes.scan(level=4, dx=128, dy=72, spp=4, depth=5, tex_size=64, lit=True),
    }[name]()
    art, img, st = _render_pair(ob, sc, abi.MODE_REFERENCE)
    # both sides draw the same Philox streams; only last-bit libm differences can split a path
    assert abs(art.Stats.TotalRays - st.total_rays) <= 1e-3 * st.total_rays
    assert abs(art.Stats.RaysLeftScene - st.rays_left_scene) <= 1e-3 * st.total_rays
    assert abs(art.Stats.RaysExceededDepth - st.rays_exceeded_depth) <= 1e-3 * st.total_rays
    assert abs(art.Stats.ShadowRays - st.shadow_rays) <= 1e-3 * max(st.shadow_rays, 1)
    diff = np.abs(art.Image[..., :3].astype(int) - img[..., :3].astype(int))
    assert (diff > 1).mean() < 5e-3, (name, (diff > 1).mean(), diff.max())
    rmse = np.sqrt(np.mean((art.Image[..., :3].astype(float) - img[..., :3].astype(float)) ** 2)) / 255.0
    assert rmse < 5e-3, rmse
    # fast mode renders the same picture
    with phys.DeviceScene(sc) as dev:
        fast = dev.render(sc.Camera[0], sc.RenderOptions, abi.MODE_FAST)
    d2 = np.abs(fast.Image[..., :3].astype(int) - img[..., :3].astype(int))
    assert (d2 > 1).mean() < 1e-2, (name, (d2 > 1).mean())


def test_render_band_sharding_sums_to_whole_image(ob):
    import ctypes as C
    import torch
    from raytrace_amd import _lib
    lib = _lib.load()
    sc = scenes.cornell(64, 80, 4, 4)
    o = sc.RenderOptions
    with phys.DeviceScene(sc) as dev:
        whole = dev.render(sc.Camera[0], o, abi.MODE_REFERENCE)
        cd = phys.camera_desc(sc.Camera[0])
        film = torch.zeros((o.Dy, o.Dx, 3), dtype=torch.float64, device="cuda")
        tot = 0
        for rank in range(3):
            ro = phys.render_opts(o, abi.MODE_REFERENCE, rank, 3)
            st = abi.Stats()
            _lib.check(lib.rt_render_film(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), C.byref(st)), lib)
            tot += st.total_rays
        torch.cuda.synchronize()
        out = np.zeros((o.Dy, o.Dx, 4), dtype=np.uint8)
        _lib.check(lib.rt_film_resolve(C.c_void_p(film.data_ptr()), o.Dx, o.Dy, o.RaysPerPixel,
                                       out.ctypes.data_as(C.POINTER(C.c_uint8)), o.Dx * 4, None), lib)
    assert tot == whole.Stats.TotalRays
    assert np.abs(out.astype(int) - whole.Image.astype(int)).max() <= 1


def test_error_codes_and_validation():
    import ctypes as C
    from raytrace_amd import _lib
    lib = _lib.load()
    sc = scenes.cornell(32, 32, 1, 2)
    with phys.DeviceScene(sc) as dev:
        img = np.zeros((32, 32, 4), dtype=np.uint8)
        cd = phys.camera_desc(sc.Camera[0])
        for bad, frag in ((dict(RaysPerPixel=0), b"RaysPerPixel"), (dict(MaxRayDepth=0), b"MaxRayDepth"), (dict(Seed=-1), b"Seed")):
            o = phys.RenderOptions(**{**dict(Seed=0, RaysPerPixel=1, MaxRayDepth=2, Dx=32, Dy=32), **bad})
            ro = phys.render_opts(o)
            rc = lib.rt_render(dev.handle, C.byref(cd), C.byref(ro), img.ctypes.data_as(C.POINTER(C.c_uint8)), 128, None)
            assert rc == abi.RT_ERR_INVALID and frag in lib.rt_last_error()
    bad_scene = phys.Scene(Camera=[], Node=[phys.Node("s", phys.Sphere((0, 0, 0), -1.0), phys.DebugUV())])
    with pytest.raises(RuntimeError, match="Sphere radius"):
        phys.DeviceScene(bad_scene)
    with pytest.raises(ValueError, match="invalid scene"):
        phys.Render(phys.Scene())

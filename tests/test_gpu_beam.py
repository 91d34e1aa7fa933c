"""The pixel-beam primary kernel (orthographic camera, one mesh, 64 or more samples per pass: the tree is walked once
per pixel for the bundle of its samples) against the per-round packet walk it replaces (RT_NO_BEAM=1) and against the
oracle.  Both kernels must find the same face at the same distance for every sample, so images and statistics are
compared for equality, not within a tolerance."""
import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load()
    return binding


def _stats(s):
    return (s.TotalRays, s.RaysLeftScene, s.RaysExceededDepth, s.ShadowRays)


def _both(sc, monkeypatch, mode=abi.MODE_FAST):
    out = []
    for no_beam in ("", "1"):
        if no_beam:
            monkeypatch.setenv("RT_NO_BEAM", "1")
        else:
            monkeypatch.delenv("RT_NO_BEAM", raising=False)
        with phys.DeviceScene(sc) as dev:
            out.append(dev.render(sc.Camera[0], sc.RenderOptions, mode))
    monkeypatch.delenv("RT_NO_BEAM", raising=False)
    return out


def grid_mesh(n=96, wobble=0.02, seed=4):
    """A nearly flat n x n sheet in the xy plane: seen edge-on, a pixel's bundle runs through hundreds of leaves."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = wobble * rng.standard_normal(X.shape)
    P = np.stack([X, Y, Z], axis=-1)
    UV = np.stack([(X + 1) / 2, (Y + 1) / 2], axis=-1)
    v = lambda i, j: np.concatenate([P[i, j], UV[i, j]], axis=-1)
    I, J = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a, b, c, d = v(I, J), v(I + 1, J), v(I + 1, J + 1), v(I, J + 1)
    return np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 5), np.stack([a, c, d], axis=-2).reshape(-1, 3, 5)])


@pytest.mark.parametrize("spp,dx,dy", [(64, 160, 96), (100, 67, 41), (256, 96, 64)])
def test_scan_beam_equals_packet_walk_and_oracle(ob, monkeypatch, spp, dx, dy):
    sc = scenes.scan(level=5, dx=dx, dy=dy, spp=spp, tex_size=64)
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image)
    assert _stats(beam.Stats) == _stats(packet.Stats)
    with ob.OracleScene(sc) as orc:
        img, st, _, _ = orc.render(sc.Camera[0], sc.RenderOptions)
    assert beam.Stats.TotalRays == st.total_rays and beam.Stats.RaysLeftScene == st.rays_left_scene
    diff = np.abs(beam.Image.astype(int) - img.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert 0.2 < (img[..., :3].max(axis=2) > 0).mean() < 0.95  # silhouette pixels are in the frame


def test_lit_scan_is_the_same_with_the_switch_either_way(monkeypatch):
    # Lambertian + point light: scenes that scatter keep the packet walk (the beam kernel with Material.Resolve in it spilled
    # too much to pay), so the switch must change nothing
    sc = scenes.scan(level=5, dx=128, dy=80, spp=64, tex_size=64, lit=True)
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image)
    assert _stats(beam.Stats) == _stats(packet.Stats)
    assert beam.Stats.ShadowRays > 0


@pytest.mark.parametrize("view", ["edge_on", "grazing", "inside", "oblique_up"])
def test_long_lists_and_odd_views(ob, monkeypatch, view):
    faces = grid_mesh()
    look = {"edge_on": ((3.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.5, 2.2),      # along the sheet: lists far beyond one instalment
            "grazing": ((3.0, 0.4, 0.12), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.6, 2.4),
            "inside": ((0.1, 0.2, 0.01), (0.9, 0.1, -0.02), (0.0, 0.0, 1.0), 0.3, 0.5),     # the rays start inside the mesh's bounds
            "oblique_up": ((1.5, 1.0, 2.5), (0.0, 0.0, 0.0), (0.3, 1.0, 0.0), 2.6, 2.6)}[view]
    cam = phys.OrthographicCamera(LookFrom=look[0], LookAt=look[1], VUp=look[2], FOVHeight=look[3], FOVWidth=look[4])
    tex = phys.TextureCheckerboard(phys.TextureUniform((0.9, 0.3, 0.2)), phys.TextureUniform((0.1, 0.5, 0.9)), 40.0)
    sc = phys.Scene(Camera=[cam], Node=[phys.Node("sheet", phys.NewMesh(faces), phys.Emitter(tex))],
                    RenderOptions=phys.RenderOptions(3, 64, 4, 120, 56))
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image), view
    assert _stats(beam.Stats) == _stats(packet.Stats), view
    assert beam.Stats.RaysLeftScene < beam.Stats.TotalRays  # something is hit
    with ob.OracleScene(sc) as orc:
        img, st, _, _ = orc.render(cam, sc.RenderOptions)
    assert beam.Stats.RaysLeftScene == st.rays_left_scene, view
    diff = np.abs(beam.Image.astype(int) - img.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, view


def test_pixels_with_long_lists_change_hands(monkeypatch):
    """A sheet seen along its plane at 512 x 512 x 512 samples: many pixels list more leaves than a list holds.  The wave that
    meets such a pixel in a batch of walks hands it on through the launch's queue of long pixels, and whichever wave draws next
    traces it as an item of its own.  Frames rendered that way (batches of the default size, of 16 and of 32 pixels), with a queue of
    five entries (when it is full a wave keeps what it meets), with every wave keeping what it meets (RT_NO_LONG_QUEUE=1) and by
    the packet walk must be the same film, number for number: a pixel's sum
    is one wave's, added in one order, whoever traces it."""
    from raytrace_amd import _lib
    lib = _lib.load()
    cam = phys.OrthographicCamera(LookFrom=(3.0, 0.0, 0.0), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0), FOVHeight=0.5, FOVWidth=2.2)
    tex = phys.TextureCheckerboard(phys.TextureUniform((0.9, 0.3, 0.2)), phys.TextureUniform((0.1, 0.5, 0.9)), 40.0)
    sc = phys.Scene(Camera=[cam], Node=[phys.Node("sheet", phys.NewMesh(grid_mesh()), phys.Emitter(tex))],
                    RenderOptions=phys.RenderOptions(3, 512, 4, 512, 512))
    films = {}
    for name, env in (("queue", {}), ("queue16", {"RT_BEAM_WALK_PX": "16"}), ("queue32", {"RT_BEAM_WALK_PX": "32"}), ("full", {"RT_BEAM_WALK_PX": "16", "RT_BEAM_LONG_CAP": "5"}),
                      ("kept", {"RT_NO_LONG_QUEUE": "1"}), ("packet", {"RT_NO_BEAM": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with phys.DeviceScene(sc) as dev:
            films[name] = _film(dev, lib, sc, frames=2)  # pilot + 510 samples from the plain order; 512 samples from the schedule
        for k in env:
            monkeypatch.delenv(k)
    assert films["queue"][1] == films["queue16"][1] == films["queue32"][1] == films["full"][1] == films["kept"][1] == films["packet"][1]
    assert films["queue"][1][1] < films["queue"][1][0]  # something is hit
    for name in ("queue16", "queue32", "full", "kept", "packet"):
        assert np.array_equal(films["queue"][0], films[name][0]), name


def test_camera_whose_rays_can_be_unsound(ob, monkeypatch):
    """An orthographic camera with a field of view near the end of the float64 range: the host cannot rule out origins that are
    not finite, so the kernel looks at every ray (render.go:147-150) -- here a share of the origins overflows.  The pixel-beam
    kernel must count and drop the same rays as the packet walk."""
    cam = phys.OrthographicCamera(LookFrom=(1e308, 0.0, 5.0), LookAt=(1e308, 0.0, 0.0), VUp=(0.0, 1.0, 0.0), FOVHeight=2.0, FOVWidth=1.7e308)
    sc = phys.Scene(Camera=[cam], Node=[phys.Node("sheet", phys.NewMesh(grid_mesh(n=16)), phys.DebugUV())], RenderOptions=phys.RenderOptions(5, 128, 3, 48, 32))
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image)
    assert _stats(beam.Stats) == _stats(packet.Stats)
    with ob.OracleScene(sc) as orc:
        _, st, _, _ = orc.render(cam, sc.RenderOptions)
    assert (beam.Stats.TotalRays, beam.Stats.RaysLeftScene) == (st.total_rays, st.rays_left_scene)
    assert 0 < st.total_rays - st.rays_left_scene  # some rays are dropped or hit
    # the same view with a sane width next to it: that one is ruled sound by the host and takes the branch-free path
    cam2 = phys.OrthographicCamera(LookFrom=(0.0, 0.0, 5.0), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 1.0, 0.0), FOVHeight=2.0, FOVWidth=1e160)
    sc2 = phys.Scene(Camera=[cam2], Node=sc.Node, RenderOptions=sc.RenderOptions)
    beam2, packet2 = _both(sc2, monkeypatch)
    assert np.array_equal(beam2.Image, packet2.Image)
    assert _stats(beam2.Stats) == _stats(packet2.Stats)


def test_tiny_and_degenerate_meshes(monkeypatch):
    # tilted: a mesh whose bounds have no thickness is never hit (AABB.hit rejects tmax <= tmin, shape.go:81-96)
    one = np.array([[[0.0, 0.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.2, 1.0, 0.0], [0.0, 1.0, 0.1, 0.0, 1.0]]])
    sliver = np.concatenate([one, one * np.array([1.0, 1e-9, 1.0, 1.0, 1.0]) + np.array([0.0, 0.0, 0.5, 0.0, 0.0])])
    cam = phys.OrthographicCamera(LookFrom=(0.3, 0.3, 5.0), LookAt=(0.3, 0.3, 0.0), VUp=(0.0, 1.0, 0.0), FOVHeight=1.5, FOVWidth=1.5)
    for faces in (one, sliver):
        sc = phys.Scene(Camera=[cam], Node=[phys.Node("m", phys.Mesh(faces), phys.DebugUV())], RenderOptions=phys.RenderOptions(1, 64, 3, 40, 40))
        beam, packet = _both(sc, monkeypatch)
        assert np.array_equal(beam.Image, packet.Image)
        assert _stats(beam.Stats) == _stats(packet.Stats)
        assert 0 < beam.Stats.RaysLeftScene < beam.Stats.TotalRays


def _film(dev, lib, sc, band=(0, 1), spp=None, frames=2):
    import ctypes as C
    import torch
    from raytrace_amd import _lib
    o = sc.RenderOptions
    film = torch.zeros((o.Dy, o.Dx, 3), dtype=torch.float64, device="cuda")
    cd = phys.camera_desc(sc.Camera[0])
    ro = phys.render_opts(o, abi.MODE_FAST, band[0], band[1], torch.cuda.current_stream().cuda_stream, 0, spp or o.RaysPerPixel)
    st = abi.Stats()
    # Twice: the first frame of a (camera, frame, share, culling) combination on a handle spends its first two samples on a
    # pilot pass that records what the tiles cost, so its sums are added up as 2 + 62 samples; the frames after it, from the
    # schedule, as one pass of 64.  The same samples either way -- but "number for number" wants the same order of additions.
    for _ in range(frames):
        film.zero_()
        _lib.check(lib.rt_render_film(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), C.byref(st)), lib)
    torch.cuda.synchronize()
    out = film.cpu().numpy()
    del film
    return out, (st.total_rays, st.rays_left_scene, st.rays_exceeded_depth, st.shadow_rays)


def test_beam_kernel_at_the_bench_size_equals_the_packet_walk_and_the_oracle(ob, monkeypatch):
    """The kernel the bench times, at the bench's size: 1 310 720 faces, 3840 x 2160, 64 samples (one launch of the beam kernel
    with 64-leaf lists, four-pixel items and instalment walks on the folded-leaf tree).  Its film -- float64 radiance sums,
    before any quantisation -- must equal the packet walk's number for number, a two-band split of it must add up to the same
    film, and on every 16th tile row and column it must equal the oracle's sums."""
    from raytrace_amd import _lib
    lib = _lib.load()
    sc = scenes.scan(level=8, dx=3840, dy=2160, spp=64, tex_size=1024)
    with phys.DeviceScene(sc) as dev:
        monkeypatch.setenv("RT_NO_PIXEL_CULL", "1")  # as in the bench's timed steps: every tile traced
        beam, beam_stats = _film(dev, lib, sc)
        monkeypatch.setenv("RT_NO_BEAM", "1")
        packet, packet_stats = _film(dev, lib, sc)
        monkeypatch.delenv("RT_NO_BEAM")
        assert beam_stats == packet_stats and beam_stats[0] == 3840 * 2160 * 64
        assert np.array_equal(beam, packet)
        del packet
        monkeypatch.delenv("RT_NO_PIXEL_CULL")
        culled, culled_stats = _film(dev, lib, sc)  # the library's default: tiles and pixels outside the outline settled by count
        assert culled_stats == beam_stats and np.array_equal(culled, beam)
        del culled
        halves = [_film(dev, lib, sc, (r, 2)) for r in (0, 1)]
        assert np.array_equal(halves[0][0] + halves[1][0], beam)  # disjoint bands: one of the two addends is zero everywhere
        assert tuple(a + b for a, b in zip(halves[0][1], halves[1][1])) == beam_stats
        del halves
    with ob.OracleScene(sc) as orc:
        img, st, _, ref = orc.render(sc.Camera[0], sc.RenderOptions, n_threads=16, tile_step=16, want_film=True)
    mask = img[..., 3] == 255  # the pixels the oracle rendered
    assert mask.sum() >= 100 * 256
    err = np.abs(beam[mask] - ref[mask]) / np.maximum(np.abs(ref[mask]), 1.0)
    assert err.max() < 1e-12, err.max()
    assert (ref[mask] > 0).any(axis=1).mean() > 0.2  # the sample is not all background


def _pinhole(origin, look_at, vup, fov_deg, aspect):
    """A PinholeCamera (camera_pinhole.go:12-34) looking from origin at look_at: image plane at unit distance."""
    o, a, up = (np.asarray(v, dtype=float) for v in (origin, look_at, vup))
    w = (o - a) / np.linalg.norm(o - a)
    u = np.cross(up, w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    hh = np.tan(np.radians(fov_deg) / 2)
    hw = hh * aspect
    llc = o - hw * u - hh * v - w
    return phys.PinholeCamera(tuple(llc), tuple(o), tuple(2 * hw * u), tuple(2 * hh * v))


PYRAMID_VIEWS = {
    # (camera, note)
    "pinhole_outside": lambda asp: _pinhole((260.0, 40.0, 90.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 52.0, asp),
    "pinhole_far_narrow": lambda asp: _pinhole((9000.0, -2500.0, 3100.0), (5.0, 0.0, -3.0), (0.0, 0.0, 1.0), 1.6, asp),   # a long lens: the point is far from the mesh frame
    "pinhole_inside": lambda asp: _pinhole((3.0, -2.0, 5.0), (60.0, 40.0, 10.0), (0.0, 0.0, 1.0), 95.0, asp),            # inside the mesh: every ray hits from within
    "pinhole_axis_aligned": lambda asp: _pinhole((0.0, 0.0, 400.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, asp),          # pixels whose directions straddle zero in x and y
    "focusable_no_aperture": lambda asp: phys.FocusableCamera(LookFrom=(210.0, -120.0, 60.0), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0),
                                                               FOVHeight=1.1, FOVWidth=1.1 * asp, Aperture=0.0, WorkingDistance=250.0),
}


@pytest.mark.parametrize("view", sorted(PYRAMID_VIEWS))
def test_beams_from_one_point_equal_the_packet_walk_and_the_oracle(ob, monkeypatch, view):
    """Pinhole cameras, and focusable cameras without aperture, start every ray at one point: their pixels' bundles are
    pyramids, walked once per pixel like the parallel bundles of an orthographic camera.  Same bar: the packet walk's image
    and statistics exactly, the oracle's to one level."""
    dx, dy, spp = 144, 96, 64
    base = scenes.scan(level=5, dx=dx, dy=dy, spp=spp, tex_size=64)
    cam = PYRAMID_VIEWS[view](dx / dy)
    sc = phys.Scene(Camera=[cam], Node=base.Node, RenderOptions=phys.RenderOptions(2, spp, 6, dx, dy))
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image), view
    assert _stats(beam.Stats) == _stats(packet.Stats), view
    with ob.OracleScene(sc) as orc:
        img, st, _, _ = orc.render(cam, sc.RenderOptions)
    assert beam.Stats.TotalRays == st.total_rays and beam.Stats.RaysLeftScene == st.rays_left_scene, view
    diff = np.abs(beam.Image.astype(int) - img.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, view
    assert (img[..., :3].max(axis=2) > 0).mean() > 0.15, view  # the mesh is in the frame


def test_pyramid_beams_on_a_sheet_seen_at_a_grazing_angle(ob, monkeypatch):
    faces = grid_mesh()
    cam = _pinhole((2.6, 0.3, 0.08), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 35.0, 120 / 56)  # lists far beyond one instalment
    tex = phys.TextureCheckerboard(phys.TextureUniform((0.9, 0.3, 0.2)), phys.TextureUniform((0.1, 0.5, 0.9)), 40.0)
    sc = phys.Scene(Camera=[cam], Node=[phys.Node("sheet", phys.NewMesh(faces), phys.Emitter(tex))], RenderOptions=phys.RenderOptions(3, 64, 4, 120, 56))
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image)
    assert _stats(beam.Stats) == _stats(packet.Stats)
    assert 0 < beam.Stats.RaysLeftScene < beam.Stats.TotalRays
    with ob.OracleScene(sc) as orc:
        img, st, _, _ = orc.render(cam, sc.RenderOptions)
    assert beam.Stats.RaysLeftScene == st.rays_left_scene
    diff = np.abs(beam.Image.astype(int) - img.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


@pytest.mark.parametrize("seed", range(24))
def test_random_meshes_and_cameras_beam_equals_packet_walk(monkeypatch, seed):
    """Random triangle soups (faces of every size and orientation, overlapping, some nearly edge-on to the rays) under random
    orthographic, pinhole and aperture-free focusable cameras, placed near, far, and inside the mesh: whatever the bundle's
    conservative tests let through, the image and the counters must be the packet walk's."""
    rng = np.random.default_rng(1000 + seed)
    n_faces = int(rng.choice([60, 400, 3000]))
    faces = scenes.random_faces(n_faces, seed=200 + seed)  # vertices in [-50, 50]^3: triangles as large as the mesh
    if seed % 2 == 1:  # small triangles: each shrunk towards its centroid
        c = faces[:, :, :3].mean(axis=1, keepdims=True)
        faces[:, :, :3] = c + 0.08 * (faces[:, :, :3] - c)
    if seed % 3 == 0:  # slivers and a few degenerate faces among them
        faces[::7, 2, :3] = faces[::7, 1, :3] + 1e-9 * rng.standard_normal((len(faces[::7]), 3))
    kind = ["ortho", "pinhole", "focusable"][seed % 3]
    dist = float(rng.choice([20.0, 160.0, 5000.0]))
    d = rng.standard_normal(3)
    d /= np.linalg.norm(d)
    if seed in (4, 7):
        d = np.array([0.0, 0.0, 1.0]) if seed == 4 else np.array([1.0, 0.0, 0.0])  # axis-aligned: zero direction components
    frm = tuple(d * dist)
    up = (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (1.0, 0.0, 0.0)
    dx, dy = 96, 64
    if kind == "ortho":
        cam = phys.OrthographicCamera(LookFrom=frm, LookAt=(0.0, 0.0, 0.0), VUp=up, FOVHeight=90.0, FOVWidth=90.0 * dx / dy)
    elif kind == "pinhole":
        cam = _pinhole(frm, (0.0, 0.0, 0.0), up, float(np.degrees(2 * np.arctan(60.0 / max(dist, 30.0)))), dx / dy)
    else:
        h = 2 * 60.0 / max(dist, 30.0)
        cam = phys.FocusableCamera(LookFrom=frm, LookAt=(0.0, 0.0, 0.0), VUp=up, FOVHeight=h, FOVWidth=h * dx / dy, Aperture=0.0, WorkingDistance=dist)
    tex = phys.TextureCheckerboard(phys.TextureUniform((0.9, 0.3, 0.2)), phys.TextureUniform((0.1, 0.5, 0.9)), 9.0)
    mat = [phys.Emitter(tex), phys.DebugUV(), phys.DebugNormal()][seed % 3 if seed < 6 else 0]
    sc = phys.Scene(Camera=[cam], Node=[phys.Node("soup", phys.NewMesh(faces), mat)], RenderOptions=phys.RenderOptions(seed, 64, 3, dx, dy))
    beam, packet = _both(sc, monkeypatch)
    assert np.array_equal(beam.Image, packet.Image), (seed, kind, dist)
    assert _stats(beam.Stats) == _stats(packet.Stats), (seed, kind, dist)
    assert beam.Stats.RaysLeftScene < beam.Stats.TotalRays, (seed, kind, dist)  # something is hit

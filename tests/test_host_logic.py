"""Host-side mirror of the reference's scene API: Validate rules, flattening into the C description,
option plumbing and the band rule.  No GPU and no library calls."""
import ctypes as C

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes, shard


def test_render_options_validate_matches_the_reference_rules():  # render.go:81-98
    ok = phys.RenderOptions(Seed=0, RaysPerPixel=1, MaxRayDepth=1, Dx=1, Dy=1)
    ok.Validate()
    for field, bad in (("Seed", -1), ("RaysPerPixel", 0), ("MaxRayDepth", 0), ("Dx", 0), ("Dy", -3)):
        o = phys.RenderOptions(Seed=0, RaysPerPixel=1, MaxRayDepth=1, Dx=1, Dy=1)
        setattr(o, field, bad)
        with pytest.raises(ValueError, match=field):
            o.Validate()


def test_shape_validate_rejects_what_the_reference_rejects():
    with pytest.raises(ValueError):
        phys.Sphere((0, 0, 0), 0.0).Validate()  # shape_sphere.go:253-269: radius must be positive
    with pytest.raises(ValueError):
        phys.Quad((0, 0, 0), (0, 0, 2), 1.0, 1.0).Validate()  # normal must be a unit vector, shape_quad.go:22-36
    with pytest.raises(ValueError):
        phys.Triangle((0, 0, 0), (1, 0, 0), (2, 0, 0)).Validate()  # collinear, shape_triangle.go:52-86
    with pytest.raises(ValueError):
        phys.Cylinder((0, 0, 0), (0, 0, 0), 1.0, 1.0).Validate()  # zero direction, shape_cylinder.go:22-36
    phys.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)).Validate()


def test_scene_validate_needs_a_camera_and_unique_node_names():  # scene.go:36-79
    s = scenes.cornell(dx=8, dy=8, spp=1)
    s.Validate()
    cams, s.Camera = s.Camera, []
    with pytest.raises(ValueError, match="no cameras"):
        s.Validate()
    s.Camera = cams
    s.Node.append(phys.Node(s.Node[0].Name, s.Node[0].Shape, s.Node[0].Material))
    with pytest.raises(ValueError, match="duplicate node name"):
        s.Validate()


def test_flatten_counts_and_sharing():
    s = scenes.cornell(dx=8, dy=8, spp=1)
    f = phys.FlatScene(s)
    d = f.desc
    assert d.n_nodes == len(s.Node)
    assert d.n_lights == len(s.Light)
    kinds = [d.nodes[i].shape_kind for i in range(d.n_nodes)]
    assert kinds.count(abi.SHAPE_QUAD) == d.n_quads
    assert kinds.count(abi.SHAPE_SPHERE) == d.n_spheres
    # one material object used by several nodes is uploaded once
    mats = {id(n.Material) for n in s.Node}
    assert d.n_materials == len(mats)
    for i in range(d.n_nodes):
        assert 0 <= d.nodes[i].material < d.n_materials


def test_flatten_mesh_faces_layout():
    s = scenes.scan(level=1, dx=8, dy=8, spp=1, tex_size=8)
    f = phys.FlatScene(s)
    d = f.desc
    assert d.n_meshes == 1 and d.n_faces == 80  # icosphere level 1
    assert C.sizeof(abi.Face) == 15 * 8  # three (x, y, z, u, v) vertices, the reference's Face (shape_mesh.go:12-16)
    assert f.faces.shape == (80, 3, 5) and f.faces.dtype == np.float64
    assert d.meshes[0].first_face == 0 and d.meshes[0].n_faces == 80


def test_render_opts_plumbing():
    o = phys.RenderOptions(Seed=7, RaysPerPixel=9, MaxRayDepth=3, Dx=20, Dy=10)
    ro = phys.render_opts(o, abi.MODE_REFERENCE, band_rank=1, band_world=2, sample_begin=4, sample_count=5)
    assert (ro.seed, ro.rays_per_pixel, ro.max_ray_depth, ro.dx, ro.dy) == (7, 9, 3, 20, 10)
    assert (ro.mode, ro.band_rank, ro.band_world, ro.sample_begin, ro.sample_count) == (abi.MODE_REFERENCE, 1, 2, 4, 5)


def test_camera_desc_kinds():
    for s, kind in ((scenes.cornell(dx=8, dy=8, spp=1), abi.CAM_PINHOLE), (scenes.scan(level=1, dx=8, dy=8, spp=1, tex_size=8), abi.CAM_ORTHOGRAPHIC),
                    (scenes.teapot(dx=8, dy=8, spp=1), abi.CAM_FOCUSABLE)):
        assert phys.camera_desc(s.Camera[0]).kind == kind


def test_scene_generators_are_deterministic():
    a = phys.FlatScene(scenes.scan(level=2, dx=8, dy=8, spp=1, tex_size=8)).faces
    b = phys.FlatScene(scenes.scan(level=2, dx=8, dy=8, spp=1, tex_size=8)).faces
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dy,world", [(1, 1), (16, 2), (100, 3), (2160, 8), (17, 4)])
def test_band_rows_partition_the_frame(dy, world):
    seen = np.concatenate([shard.band_rows(dy, r, world) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(dy))
    for r in range(world):
        rows = shard.band_rows(dy, r, world)
        assert all(shard.owns_row(int(y), r, world) for y in rows)
        assert all((int(y) // shard.BAND_ROWS) % world == r for y in rows)


def test_animated_camera_resolves_on_the_host():  # camera_animated.go:53-131
    import math

    def orbit(u):
        a = 2 * math.pi * u
        return phys.OrthographicCamera(LookFrom=(10 * math.cos(a), 10 * math.sin(a), 5.0), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0),
                                       FOVHeight=4.0, FOVWidth=4.0)

    ac = phys.AnimatedCamera(orbit, U=1.25, Period=8.0)
    ac.Validate()
    want = phys.camera_desc(orbit(0.25))
    got = phys.camera_desc(ac)  # U wraps into [0, 1) at cast time
    assert got.kind == abi.CAM_ORTHOGRAPHIC and tuple(got.look_from) == tuple(want.look_from)
    assert phys.camera_desc(ac.WithTime(2.0)).look_from[0] == want.look_from[0]  # 2 s of an 8 s period = a quarter turn
    assert abs(ac.Advance(4.0).U - 1.75) < 1e-15
    assert len(ac.Frames(5)) == 5 and ac.Frames(0) is None
    with pytest.raises(ValueError, match="Build is nil"):
        phys.AnimatedCamera(None).Validate()
    with pytest.raises(ValueError, match="Period is zero"):
        phys.AnimatedCamera(orbit).WithTime(1.0)


def test_three_instruction_division_is_the_ieee_quotient():
    """The kernels divide by Dx, Dy and 65535 as q0 = n*r, q = fma(fma(-q0, d, n), r, q0) with r = 1/d (div_by in
    dev_math.h).  Emulated here with exact rationals (a fused multiply-add rounds once): the result must be the correctly
    rounded quotient, i.e. what the reference's plain division gives (render.go:202-210, texture_image.go:183-187)."""
    import random
    from fractions import Fraction as F

    def div_by(n, d):
        r = 1.0 / d
        q0 = n * r
        rem = float(F(n) - F(q0) * d)
        return float(F(q0) + F(rem) * F(r))

    for c in range(65536):
        assert div_by(float(c), 65535) == c / 65535.0
    rng = random.Random(7)
    for d in (1, 3, 41, 67, 96, 1080, 1920, 2160, 3840, 4096, 65535, 1000003):
        for _ in range(3000):
            n = float(rng.randrange(0, d)) + rng.getrandbits(53) / 2.0 ** 53
            assert div_by(n, d) == n / d, (n, d)
        for x in range(min(d, 500)):
            assert div_by(x + 0.5, d) == (x + 0.5) / d


def test_bench_gpus_n_starts_its_own_ranks_before_touching_the_gpu(monkeypatch):
    """`bench.py --gpus N` as one process (no WORLD_SIZE): the ranks are started through torch.distributed.run on 127.0.0.1, with the
    caller's arguments, and the parent takes the child's exit code -- checked here on the command it builds."""
    import subprocess
    import sys
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    a = types.SimpleNamespace(gpus=4)
    assert bench.self_launch(a) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_stored_counters_are_tied_to_the_device_sources():
    import json
    import os
    import bench
    h = bench.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and h == bench.source_hash()
    stored = json.load(open(os.path.join(bench.ROOT, "profiles", "counters.json")))
    assert "csrc_sha16" in stored and stored["kernel"].startswith("k_beam")
    # the floor the roofline prices: every stage named once, positive, and below what the kernel was measured to execute
    stages = [s for s, _, _, _ in bench.VALU_FLOOR]
    assert len(stages) == len(set(stages)) and all(n > 0 for _, _, n, _ in bench.VALU_FLOOR)
    assert {who for _, who, _, _ in bench.VALU_FLOOR} == {"every ray", "listed", "hit"}
    worst = sum(n for _, _, n, _ in bench.VALU_FLOOR)  # a round in which every ray hits
    assert worst < stored["valu_insts_per_wave_of_64_rays"] * 1.5

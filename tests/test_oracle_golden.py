"""Pins the CPU oracle against the reference's own known-answer tests.

Fixtures in tests/golden/*.json are transcribed by hand from the reference's Go
test tables (file:line in each fixture's "source").  Where a reference test
table and the reference code disagree the code wins (it is what renders); those
cases carry both values.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import binding as ob
from raytrace_amd import phys

G = os.path.join(os.path.dirname(__file__), "golden")
EPS = 1e-6
MAXF = 1.7976931348623157e308


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def one_shape_scene(shape, material=None):
    return phys.Scene(Camera=[], Node=[phys.Node("s", shape, material or phys.DebugNormal())])


def collide(shape, o, d, tmin, tmax):
    with ob.OracleScene(one_shape_scene(shape)) as sc:
        return sc.intersect([o], [d], tmin, tmax)[0]


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def test_triangle_collide_tables():
    g = load("triangle_collide.json")
    n = 0
    for grp in g["groups"]:
        tri = phys.Triangle(*grp["triangle"])
        for c in grp["cases"]:
            h = collide(tri, c["o"], c["d"], 0.0, MAXF)
            want = c.get("code_hit", c["expect_hit"])
            assert (h["node"] >= 0) == want, c["name"]
            if want:
                assert np.allclose(h["at"], c["point"], atol=EPS, rtol=0), c["name"]
                assert np.allclose(h["normal"], c["normal"], atol=EPS, rtol=0), c["name"]
            n += 1
    assert n == 30


def test_triangle_tmin_zero_cases_are_hits_at_negative_zero():
    # shape_triangle_test.go:393-411 expects misses, but t = f*(e2.q) = -1*0 = -0.0 and
    # `-0.0 < 0` is false in Triangle.Collide (shape_triangle.go:128-131).
    tri = phys.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0))
    h = collide(tri, (0.5, 0.5, 0), (0, 0, 1), 0.0, MAXF)
    assert h["node"] == 0 and h["t"] == 0.0 and math.copysign(1.0, h["t"]) == -1.0
    # with the renderer's tmin = eps they are misses
    assert collide(tri, (0.5, 0.5, 0), (0, 0, 1), EPS, MAXF)["node"] == -1


def test_triangle_bounds_and_validate():
    g = load("triangle_collide.json")
    b = g["bounds"]
    with ob.OracleScene(one_shape_scene(phys.Triangle(*b["triangle"]))) as sc:
        mn, mx = sc.node_bounds(0)
    assert np.allclose(mn, b["min"], atol=EPS) and np.allclose(mx, b["max"], atol=EPS)
    for t in g["validate"]["valid"]:
        phys.Triangle(*t).Validate()
    for t in g["validate"]["invalid"]:
        with pytest.raises(ValueError):
            phys.Triangle(*t).Validate()


def test_triangle_fuzz_properties_on_seed_corpus_and_random():
    # FuzzTriangleCollide properties (shape_triangle_test.go:101-137)
    g = load("triangle_collide.json")["fuzz_seed_corpus"]
    cases = [(t, r["o"], r["d"]) for t in g["triangles"] for r in g["rays"]]
    rng = np.random.default_rng(5)
    for _ in range(300):
        t = (rng.random((3, 3)) * 4 - 2).tolist()
        cases.append((t, (rng.random(3) * 6 - 3).tolist(), (rng.random(3) * 2 - 1).tolist()))
    hits = 0
    for t, o, d in cases:
        tri = phys.Triangle(*t)
        try:
            tri.Validate()
        except ValueError:
            continue
        d = unit(d)
        h = collide(tri, o, d, 0.0, MAXF)
        if h["node"] < 0:
            continue
        hits += 1
        p0, p1, p2 = (np.array(x, dtype=np.float64) for x in t)
        n = unit(np.cross(p1 - p0, p2 - p0))
        assert abs(n @ (h["at"] - p0)) <= EPS
        u, v = h["uv"]
        assert -EPS <= u <= 1 + EPS and -EPS <= v <= 1 + EPS and u + v <= 1 + EPS
        assert np.allclose(h["normal"], n, atol=EPS, rtol=0)
        assert np.allclose(h["at"], np.array(o) + d * h["t"], atol=EPS, rtol=0)
    assert hits >= 10


def test_cylinder_tables():
    g = load("cylinder.json")
    cyl = phys.Cylinder(g["cylinder"]["origin"], g["cylinder"]["direction"], g["cylinder"]["radius"], g["cylinder"]["height"])
    for c in g["cases"]:
        h = collide(cyl, c["o"], unit(c["d"]), 0.0001, MAXF)
        assert (h["node"] >= 0) == c["hit"], c["name"]
        if c["hit"]:
            assert np.allclose(h["at"], c["point"], atol=EPS, rtol=0), c["name"]
            assert np.allclose(h["normal"], c["normal"], atol=EPS, rtol=0), c["name"]
    for b in g["bounds"]:
        with ob.OracleScene(one_shape_scene(phys.Cylinder(b["origin"], b["direction"], b["radius"], b["height"]))) as sc:
            mn, mx = sc.node_bounds(0)
        assert np.allclose(mn, b["min"], atol=EPS) and np.allclose(mx, b["max"], atol=EPS), b["name"]


def test_quad_table():
    g = load("quad.json")
    q = phys.Quad(g["quad"]["center"], g["quad"]["normal"], g["quad"]["width"], g["quad"]["height"])
    for c in g["cases"]:
        h = collide(q, c["o"], c["d"], EPS, 1000.0)
        assert (h["node"] >= 0) == c["hit"], c["name"]
        if c["hit"]:
            assert np.allclose(h["at"], c["point"], atol=EPS, rtol=0)
            assert np.allclose(h["normal"], c["normal"], atol=EPS, rtol=0)
            assert np.allclose(h["uv"], c.get("code_uv", c["expected_uv"]), atol=EPS, rtol=0), c["name"]


def test_aabb_intersects():
    for c in load("misc.json")["aabb_intersects"]["cases"]:
        assert ob.aabb_intersects(c["a"][0], c["a"][1], c["b"][0], c["b"][1]) == c["intersect"]


def test_aabb_hit_semantics():
    # shape.go:81-96: strict `tmax <= tmin` rejection, so a zero-thickness box is never hit,
    # and Go's NaN-propagating Max/Min let a ray lying in a slab plane through.
    assert ob.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), 1e-6, MAXF)
    assert not ob.aabb_hit((0, 0, 0), (1, 1, 0), (0.5, 0.5, -1), (0, 0, 1), 1e-6, MAXF)
    assert not ob.aabb_hit((0, 0, 0), (1, 1, 1), (2, 0.5, -1), (0, 0, 1), 1e-6, MAXF)
    assert ob.aabb_hit((0, 0, 0), (1, 1, 1), (0.0, 0.5, -1), (0, 0, 1), 1e-6, MAXF)  # 0*inf = NaN on x
    assert not ob.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), 1e-6, 0.5)


def test_r3_examples_exact():
    g = load("misc.json")["r3_examples"]
    assert ob.r3(7, ob.r3(1, (1, 2, 3), (4, 5, 6))) == g["distance_1_2_3__4_5_6"]
    disp = ob.r3(3, ob.r3(8, (1, 1, 0)), s=10.0 * 0.5)
    assert (ob.r3(0, (0, 0, 0), disp) == np.array(g["moved_position"])).all()
    assert (ob.r3(9, (0, 0, 0), (10, 10, 10), s=0.5) == np.array(g["lerp_half"])).all()
    ang = math.acos(ob.r3(5, (1, 0, 0), (0, 1, 0)) / (ob.r3(7, (1, 0, 0)) * ob.r3(7, (0, 1, 0)))) * (180 / math.pi)
    assert f"{ang:.2f}" == g["angle_degrees_2dp"]
    inc = ob.r3(8, (1, -1, 0))
    refl = ob.r3(1, inc, ob.r3(3, (0, 1, 0), s=2 * ob.r3(5, inc, (0, 1, 0))))
    assert (refl == np.array(g["reflection"])).all()
    vec = ob.r3(1, (5, 10, 5), (0, 0, 0))
    proj = ob.r3(1, (5, 10, 5), ob.r3(3, (0, 1, 0), s=ob.r3(5, vec, (0, 1, 0))))
    assert (proj == np.array(g["projected_point"])).all()
    # Go's math.Cos is its own (Cephes-derived) routine: cos(pi/2) prints 6.123233995736757e-17 there and
    # 6.123233995736766e-17 with glibc, so the reference's value is fed in rather than recomputed here.
    c, s = 6.123233995736757e-17, 1.0
    v, axis = (1, 0, 0), (0, 0, 1)
    rot = ob.r3(0, ob.r3(0, ob.r3(3, v, s=c), ob.r3(3, ob.r3(6, axis, v), s=s)), ob.r3(3, axis, s=ob.r3(5, axis, v) * (1 - c)))
    assert (rot == np.array(g["rotated_vector"])).all()
    assert (ob.r3(3, ob.r3(8, (3, 4, 0)), s=10.0) == np.array(g["scaled_vector"])).all()
    # r3/vec.go:91-97 and :62-70
    assert (ob.r3(8, (0, 0, 0)) == 0).all()
    assert (ob.r3(9, (0, 0, 0), (2, 2, 2), s=7.0) == 2).all()


def test_microfacet_against_reference_test_expectations():
    from microfacet_cases import check_microfacet
    check_microfacet(ob.microfacet)


def test_microfacet_legacy_known_answer():
    # microfacet_test.go:484 keeps a commented known answer, Evaluate(R=0.3, F0=0.04, wo=wi=n=z) =
    # 0.035367765043112884, printed by an older revision whose denominator was 4*cos*cos + 1e-8
    # (the current code adds eps = 1e-6, microfacet.go:100).  Undoing that one constant reproduces it.
    want = load("misc.json")["microfacet"]["legacy_evaluate_R0.3_F0.04_all_z_without_eps_in_denominator"]
    got = ob.microfacet(4, 0.3, (0.04, 0.04, 0.04), (0, 0, 1), (0, 0, 1), (0, 0, 1))[0]
    assert abs(got * (4 + EPS) / (4 + 1e-8) - want) < 2e-17


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert ob.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert ob.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert ob.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_random_stream_distribution():
    x = ob.random_stream(3, 17, 2, 1, 20000)
    assert x.min() >= 0 and x.max() < 1 and abs(x.mean() - 0.5) < 0.01
    assert not np.array_equal(x[:100], ob.random_stream(3, 18, 2, 1, 100))
    assert not np.array_equal(x[:100], ob.random_stream(3, 17, 2, 2, 100))
    # rand.go:26-39, 62-75, 79-105
    for i in range(50):
        assert np.linalg.norm(ob.sample(0, 1, i, 0)) < 1.0
        p = ob.sample(1, 1, i, 0)
        assert p[2] == 0 and p @ p < 1.0
        nrm = unit((0.3, -0.5, 0.8))
        d = ob.sample(2, 1, i, 0, nrm)
        assert abs(np.linalg.norm(d) - 1) < 1e-12 and d @ nrm >= 0


def test_r3_table_tests():
    """r3/vec_test.go: TestVecSub, Mul, Muls, Divs, Dot, Cross, Length, Unit, Lerp (clamped), Clip (transcribed)."""
    from oracle import binding as ob
    v1, v2 = (1, 2, 3), (4, 5, 6)
    assert np.array_equal(ob.r3(1, v1, v2), [-3, -3, -3])
    assert np.array_equal(ob.r3(2, v1, v2), [4, 10, 18])
    assert np.array_equal(ob.r3(3, v1, s=2.0), [2, 4, 6])
    assert np.array_equal(ob.r3(4, (2, 4, 6), s=2.0), [1, 2, 3])
    assert np.isinf(ob.r3(4, (2, 4, 6), s=0.0)).all()  # Divs(0): IEEE infinities, as in Go
    assert ob.r3(5, v1, v2) == 32.0
    assert np.array_equal(ob.r3(6, v1, v2), [-3, 6, -3])
    assert ob.r3(7, (3, 4, 0)) == 5.0
    assert np.allclose(ob.r3(8, (3, 4, 0)), [0.6, 0.8, 0], atol=1e-6, rtol=0)
    assert np.array_equal(ob.r3(8, (0, 0, 0)), [0, 0, 0])
    for t, want in ((0, 0), (0.5, 5), (1, 10), (-0.5, 0), (1.5, 10)):  # TestVecLerp: the parameter is clamped to [0, 1]
        assert np.array_equal(ob.r3(9, (0, 0, 0), (10, 10, 10), s=t), [want] * 3)
    assert np.array_equal(ob.r3(10, (-2, 0, 2), s=-1.0, t=1.0), [-1, 0, 1])

"""Materials and the Beckmann BRDF per hit, on the GPU: the reference's own microfacet tables
(phys/microfacet_test.go:12-443) run against the HIP functions through rt_microfacet_batch, and Material.Resolve +
ComputeDirectLighting of every material kind for 10^5 random hits through rt_shade_batch against the oracle
(oracle.cpp resolve / direct_lighting, the restatement of material_*.go).

Bars: counts (children, draws, shadow rays) exact.  Values bit-exact wherever the arithmetic is +, -, *, /, sqrt
(Metal and Dielectric scatter without roughness, Lambertian direct light, every emission); where a libm function
is involved (sin / cos in CosineWeightedHemisphere, pow in Schlick, exp in the Beckmann D, sin in the checkerboard)
the device's and glibc's results may differ in the last bits: 1e-12 relative there."""
import math

import numpy as np
import pytest

from microfacet_cases import EPS, check_microfacet
from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load()
    return binding


def _gpu_mf(which, roughness, f0, a, b=(0, 0, 0), c=(0, 0, 0), d=(0, 0, 0)):
    out = phys.microfacet_batch(which, roughness, f0, np.array([[a, b, c, d]], dtype=np.float64))[0]
    return float(out[0]) if which < 3 else out


def test_reference_microfacet_tables_on_the_gpu():
    check_microfacet(_gpu_mf)
    # the legacy known answer (microfacet_test.go:484), see tests/test_oracle_golden.py
    got = _gpu_mf(abi.MF_EVALUATE, 0.3, (0.04, 0.04, 0.04), (0, 0, 1), (0, 0, 1), (0, 0, 1))[0]
    assert abs(got * (4 + EPS) / (4 + 1e-8) - 0.035367765043112884) < 2e-17


@pytest.mark.parametrize("which", [abi.MF_D, abi.MF_G1, abi.MF_G, abi.MF_F, abi.MF_EVALUATE])
def test_microfacet_functions_match_the_oracle_on_random_arguments(ob, which):
    rng = np.random.default_rng(100 + which)
    n = 20000
    args = rng.normal(size=(n, 4, 3))
    args[: n // 2] /= np.linalg.norm(args[: n // 2], axis=2, keepdims=True)  # half of them unit vectors, half not
    rough = rng.random(n)
    rough[:50] = 0.0  # microfacet.go:29 clamps D's roughness; G1 divides by the raw value
    f0 = rng.random((n, 3))
    got = phys.microfacet_batch(which, rough, f0, args)
    want = np.array([np.atleast_1d(ob.microfacet(which, rough[i], f0[i], *args[i])) for i in range(0, n, 7)])
    got = got[::7, : want.shape[1]]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-12, atol=1e-300)
    assert (got[ok] == want[ok]).mean() > 0.5  # and mostly to the bit


def _hits(ob, sc, n, seed):
    """Random hits on the scene: (record array of rt_shade_in without material, node materials)."""
    rng = np.random.default_rng(seed)
    flat = phys.FlatScene(sc)
    lo, hi = np.array([-50e6, -30e6, -4e6]), np.array([50e6, 20e6, 4e6])  # the sphere grid of example/material, nm
    recs, have = [], 0
    with ob.OracleScene(flat) as orc:
        while have < n:
            m = 2 * n
            o = lo + rng.random((m, 3)) * (hi - lo)
            o[:, 2] = 300e6 * np.where(rng.random(m) < 0.8, 1.0, -1.0)
            target = lo + rng.random((m, 3)) * (hi - lo)
            d = target - o
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            h = orc.intersect(o, d)
            ok = h["node"] >= 0
            inside = ok & (rng.random(m) < 0.15)  # some rays go on from just inside the sphere they hit: hits from inside (Dielectric's other branch)
            h2 = orc.intersect(h["at"][inside] + d[inside] * 1e3, d[inside])
            ok2 = h2["node"] >= 0
            for hh, dd in ((h[ok & ~inside], d[ok & ~inside]), (h2[ok2], d[inside][ok2])):
                r = np.zeros(len(hh), dtype=abi.SHADE_IN_DTYPE)
                r["node"], r["t"], r["at"], r["normal"], r["uv"], r["d"] = hh["node"], hh["t"], hh["at"], hh["normal"], hh["uv"], dd
                recs.append(r)
                have += len(r)
    r = np.concatenate(recs)[:n]
    assert len(r) == n
    r["radiance"] = rng.random((n, 3))
    r["pixel"] = rng.integers(0, 1 << 20, n)
    r["sample"] = rng.integers(0, 4096, n)
    r["path"] = rng.integers(1, 64, n)
    r["draws"] = rng.integers(0, 9, n)
    node_mat = np.array([flat.desc.nodes[i].material for i in range(flat.desc.n_nodes)])
    mat_kind = np.array([flat.desc.materials[i].kind for i in range(flat.desc.n_materials)])
    return flat, r, node_mat, mat_kind


def _compare(got, want, exact_fields, close_fields):
    for f in ("n_children", "draws_after", "shadow_rays"):
        assert np.array_equal(got[f], want[f]), f
    for f in exact_fields:
        assert np.array_equal(got[f], want[f]), f
    for f in close_fields:
        assert np.allclose(got[f], want[f], rtol=1e-12, atol=1e-300), (f, np.abs(got[f] - want[f]).max())


@pytest.mark.parametrize("mode", [abi.MODE_REFERENCE, abi.MODE_FAST])
def test_resolve_and_direct_lighting_per_hit_match_the_oracle(ob, mode):
    n = 100000
    sc = scenes.material(dx=64, dy=64, spp=1, tex_size=(32, 32))
    sc.Light.append(phys.PointLight((-300.0 * phys.MM, 200.0 * phys.MM, 150.0 * phys.MM), (0.2, 0.1, 0.4)))
    flat, hits, node_mat, mat_kind = _hits(ob, sc, n, seed=7)
    names = {abi.MAT_LAMBERTIAN: "Lambertian", abi.MAT_METAL: "Metal", abi.MAT_DIELECTRIC: "Dielectric", abi.MAT_EMITTER: "Emitter",
             abi.MAT_DEBUG_UV: "DebugUV", abi.MAT_DEBUG_NORMAL: "DebugNormal"}
    with phys.DeviceScene(sc) as dev, ob.OracleScene(flat) as orc:
        # 1. every hit with the material of the node it lies on
        hits["material"] = node_mat[hits["node"]]
        got, want = dev.shade(hits, seed=3, mode=mode), orc.shade(hits, seed=3)
        _compare(got, want, ["emission"] if mode == abi.MODE_REFERENCE else [], ["emission", "dir", "rad", "direct"])
        assert want["shadow_rays"].sum() > n  # two lights, lit materials on most nodes
        assert 0.02 < (want["direct"] == 0).all(axis=1).mean() < 0.98  # some hits in shadow, some lit
        # 2. every material kind on all n hits, whatever node they lie on
        for kind, name in names.items():
            mats = np.flatnonzero(mat_kind == kind)
            assert len(mats), name
            hits["material"] = mats[np.arange(n) % len(mats)]
            got, want = dev.shade(hits, seed=11, mode=mode), orc.shade(hits, seed=11)
            exact = {abi.MAT_LAMBERTIAN: ["emission", "rad"], abi.MAT_EMITTER: ["emission", "dir", "rad", "direct"],
                     abi.MAT_DEBUG_UV: ["emission", "dir", "rad", "direct"], abi.MAT_DEBUG_NORMAL: ["emission", "dir", "rad", "direct"]}.get(kind, [])
            _compare(got, want, exact, ["emission", "dir", "rad", "direct"])
            if kind == abi.MAT_DIELECTRIC:
                assert (want["n_children"] == 2).mean() > 0.5 and (want["n_children"] == 1).any()  # refraction + reflection, and total reflection
            if kind == abi.MAT_METAL:
                assert (want["n_children"] == 0).any() and (want["draws_after"] - hits["draws"] >= 3).all()
            if kind in (abi.MAT_METAL, abi.MAT_DIELECTRIC):  # no libm in Resolve unless the lobe is rough: mostly to the bit
                assert (got["dir"] == want["dir"]).all(axis=(1, 2)).mean() > 0.9


def test_shade_batch_rejects_bad_indices():
    sc = scenes.cornell(dx=8, dy=8, spp=1)
    with phys.DeviceScene(sc) as dev:
        h = np.zeros(1, dtype=abi.SHADE_IN_DTYPE)
        h["material"] = 99
        with pytest.raises(RuntimeError, match="material index"):
            dev.shade(h)
        h["material"], h["node"] = 0, 99
        with pytest.raises(RuntimeError, match="node index"):
            dev.shade(h)

"""The expectations of the reference's microfacet tests (phys/microfacet_test.go:12-443), written once and run against
the CPU oracle (tests/test_oracle_golden.py) and against the HIP functions (tests/test_gpu_materials.py)."""
import math

import numpy as np

EPS = 1e-6  # the tests' own tolerance (microfacet_test.go: const epsilon)


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def _closed_D(r, ct):
    r = max(r, EPS)
    return math.exp((ct ** 2 - 1) / (r ** 2 * ct ** 2)) / (math.pi * r ** 2 * ct ** 2 * ct ** 2)


def check_microfacet(mf):
    """mf(which, roughness, f0, a, b, c, d): 0 = D(h, n), 1 = G1(v, n, h), 2 = G(wo, wi, n, h) -> float; 3 = F(wo, h), 4 = Evaluate(wo, wi, n) -> 3 values."""
    F0 = (0.04, 0.04, 0.04)
    n = (0.0, 0.0, 1.0)
    # TestD (microfacet_test.go:12-46)
    assert abs(mf(0, 0.5, F0, n, n) - _closed_D(0.5, 1.0)) < EPS
    assert abs(mf(0, 0.5, F0, (1, 0, 0), n)) < EPS
    assert not math.isnan(mf(0, 0.0, F0, n, n))
    # TestG1 (:49-94)
    assert abs(mf(1, 0.5, F0, n, n, n) - 1.0) < EPS
    th = math.atan(1.25)
    assert abs(mf(1, 0.5, F0, unit((math.sin(th), 0, math.cos(th))), n, n) - 1.0) < EPS
    th = math.atan(2.0)
    a = 1 / (0.5 * (math.sqrt(1 - math.cos(th) ** 2) / math.cos(th)))
    want = (3.535 * a + 2.181 * a * a) / (1 + 2.276 * a + 2.577 * a * a)
    assert abs(mf(1, 0.5, F0, unit((math.sin(th), 0, math.cos(th))), n, n) - want) < EPS
    assert abs(mf(1, 0.5, F0, (1, 0, 0), n, n)) < EPS
    # TestG (:97-127)
    assert abs(mf(2, 0.5, F0, n, n, n, n) - 1.0) < EPS
    wo = unit((math.sin(math.pi / 4), 0, math.cos(math.pi / 4)))
    g1 = mf(1, 0.5, F0, wo, n, wo)
    assert abs(mf(2, 0.5, F0, wo, wo, n, wo) - g1 * g1) < EPS
    # TestF (:130-166)
    assert np.allclose(mf(3, 0.5, F0, n, n), F0, atol=EPS)
    assert np.allclose(mf(3, 0.5, F0, (1, 0, 0), (0, 1, 0)), 1.0, atol=EPS)
    f45 = 0.04 + 0.96 * (1 - math.cos(math.pi / 4)) ** 5
    assert np.allclose(mf(3, 0.5, F0, wo, wo), f45, atol=1e-2)
    # TestEvaluate (:169-243)
    want = 0.04 * _closed_D(0.5, 1.0) * 1.0 / 4
    assert np.allclose(mf(4, 0.5, F0, n, n, n), want, atol=EPS)
    D = mf(0, 0.5, F0, wo, n)
    Gv = mf(2, 0.5, F0, wo, wo, n, wo)
    Fv = mf(3, 0.5, F0, wo, wo)
    denom = 4 * max(0, wo[2]) * max(0, wo[2])
    assert np.allclose(mf(4, 0.5, F0, wo, wo, n), Fv * (D * Gv / denom), atol=EPS)
    assert np.allclose(mf(4, 0.5, F0, wo, (0, 0, -1), n), 0.0, atol=EPS)
    smooth = mf(4, 0.0, F0, n, n, n)
    assert not np.isnan(smooth).any()
    assert np.allclose(smooth, 0.04 * _closed_D(0.0, 1.0) / 4, atol=1e3)
    w = unit((1, 0, 1))
    D = mf(0, 1.0, F0, w, n)
    Gv = mf(2, 1.0, F0, w, w, n, w)
    Fv = mf(3, 1.0, F0, w, w)
    assert np.allclose(mf(4, 1.0, F0, w, w, n), Fv * (D * Gv / (4 * w[2] * w[2])), atol=EPS)
    # TestEvaluatePhysicalProperties (:336-365)
    wo = unit((math.sin(math.pi / 6), 0, math.cos(math.pi / 6)))
    wi = unit((math.sin(math.pi / 3), 0, math.cos(math.pi / 3)))
    s1, s2 = mf(4, 0.5, F0, wo, wi, n), mf(4, 0.5, F0, wi, wo, n)
    assert (s1 >= 0).all() and np.allclose(s1, s2, atol=EPS)
    Fh = mf(3, 0.5, F0, wo, unit(wo + wi))
    assert (s1 <= Fh + EPS).all()
    # TestZeroRoughnessAndF0 (:379-395), TestNonNormalizedVectors (:398-423), TestDFunctionProperties (:426-443)
    assert np.allclose(mf(4, 0.0, (0, 0, 0), n, n, n), 0.0, atol=EPS)
    assert np.allclose(mf(4, 0.5, F0, (0, 0, 2), (0, 0, 2), n), mf(4, 0.5, F0, n, n, n), atol=EPS)
    assert abs(mf(0, 0.3, F0, unit((1, 1, 1)), n) - mf(0, 0.3, F0, unit((-1, -1, 1)), n)) < EPS


